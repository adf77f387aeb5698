// Exhaustive check (run on the GPU box): for d = r*r, r = 3..10, is  q0 = x*rcp; e = fma(-d, q0, x); q = fma(e, rcp, q0)
// equal to the correctly rounded x / d for EVERY float x the blobCenter stage can produce, i.e. every (float)n with n an
// int32 in [-2^27, 2^27]?  (box sums are bounded by r*r * 3*255^2 < 2^25.)   Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(int d, unsigned long long* bad, int* first) {
	const float fd = (float)d, rcp = 1.0f / fd;
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x - (1ll << 27);
	if(i > (1ll << 27)) return;
	const float x = (float)(int)i;
	const float ref = x / fd;
	const float q0 = x * rcp;
	const float e = __builtin_fmaf(-fd, q0, x);
	const float q = __builtin_fmaf(e, rcp, q0);
	if(q != ref) {
		atomicAdd(bad, 1ull);
		atomicMin(first, (int)(i < 0 ? -i : i));
	}
}
int main() {
	unsigned long long* bad;
	int* first;
	hipMalloc(&bad, 8);
	hipMalloc(&first, 4);
	int rc = 0;
	for(int r = 3; r <= 10; r++) {
		hipMemset(bad, 0, 8);
		int big = 0x7fffffff;
		hipMemcpy(first, &big, 4, hipMemcpyHostToDevice);
		const long long n = (1ll << 28) + 1;
		k<<<(unsigned)((n + 255) / 256), 256>>>(r * r, bad, first);
		unsigned long long hb;
		int hf;
		hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
		hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
		printf("r=%d d=%d mismatches=%llu first|x|=%d\n", r, r * r, hb, hb ? hf : -1);
		if(hb) rc = 1;
	}
	return rc;
}
