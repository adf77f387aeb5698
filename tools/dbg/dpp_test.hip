// cross-lane primitive check on the real device: DPP wave shifts and ds_bpermute, on loaded values
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int* in, int* out) {
	const int lane = threadIdx.x;
	const int e = in[lane];
	out[lane] = __builtin_amdgcn_update_dpp(-1, e, 0x130, 0xf, 0xf, false);        // wave_shl:1
	out[64 + lane] = __builtin_amdgcn_update_dpp(-1, e, 0x138, 0xf, 0xf, false);   // wave_shr:1
	out[128 + lane] = __builtin_amdgcn_ds_bpermute(4 * (lane + 5), e);
	out[192 + lane] = __builtin_amdgcn_ds_bpermute(4 * (lane - 5), e);
	int ph = 4 * ((lane + 5) & 63);
	asm volatile("" : "+v"(ph));
	out[256 + lane] = __builtin_amdgcn_ds_bpermute(ph, e);
	out[320 + lane] = __builtin_amdgcn_update_dpp(-1, e, 0x134, 0xf, 0xf, false);  // wave_rol:1
	out[384 + lane] = __builtin_amdgcn_update_dpp(-1, e, 0x101, 0xf, 0xf, false);  // row_shl:1
	out[448 + lane] = __builtin_amdgcn_update_dpp(-1, e, 0x111, 0xf, 0xf, false);  // row_shr:1
	// what the DPP combiner does to  a - dpp(e)  and  dpp(e) - a   (a = another loaded value)
	const int a = in[64 + lane];
	out[512 + lane] = a - __builtin_amdgcn_update_dpp(0, e, 0x130, 0xf, 0xf, false);   // expected: a[l] - e[l+1]
	out[576 + lane] = __builtin_amdgcn_update_dpp(0, e, 0x138, 0xf, 0xf, false) - a;   // expected: e[l-1] - a[l]
	out[640 + lane] = a - __builtin_amdgcn_update_dpp(0, e, 0x111, 0xf, 0xf, false);   // row_shr:1, expected a[l] - e[l-1]
}
int main() {
	int *din, *dout, hin[128], hout[704];
	for(int i = 0; i < 64; i++) { hin[i] = 100 + i; hin[64 + i] = 10000 + 100 * i; }
	hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(hout));
	hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
	k<<<1, 64>>>(din, dout);
	hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
	const char* names[11] = {"wave_shl:1", "wave_shr:1", "bperm +5 (folded)", "bperm -5", "bperm +5 (opaque)", "wave_rol:1", "row_shl:1", "row_shr:1", "a - shl(e)", "shr(e) - a", "a - row_shr(e)"};
	for(int r = 0; r < 11; r++) {
		printf("%-18s:", names[r]);
		for(int i = 0; i < 64; i++) printf(" %d", hout[64 * r + i]);
		printf("\n");
	}
	return 0;
}
