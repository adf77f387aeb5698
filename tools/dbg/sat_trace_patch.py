"""Adds phase timestamps (s_memrealtime, thread 0 of every workgroup, first 16 tiles) to k_sat_fast2 and two debug exports
(vp_dbg_trace_clear / vp_dbg_trace_copy) -- a TEMPORARY instrumentation for tools/dbg/sattrace.py.  Apply, `make hip`, run sattrace.py on the
GPU, then `git checkout vision-processor_amd/csrc/vp_sat_tiled.hip`.  Never commit the patched source."""
import os
p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "vision-processor_amd", "csrc", "vp_sat_tiled.hip")
s = open(p).read()
k = s.index("template<int O, int WPC, bool GD, bool RAG>\n__global__ __launch_bounds__(256, WPC) void k_sat_fast2(")
head, body = s[:k], s[k:]


def rep(a, b):
    global body
    assert body.count(a) >= 1, a[:60]
    body = body.replace(a, b, 1)


head = head.replace("	int single;     // k_sat_fast2:", "	unsigned long long* trace;\n	int single;     // k_sat_fast2:", 1)
body = "#define TR(P) do { if(tid == 0 && a.trace && trn < 16) a.trace[((size_t)blockIdx.x * 16 + trn) * 8 + (P)] = __builtin_amdgcn_s_memrealtime(); } while(0)\n" + body
rep("	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;\n	const int w = a.w, h = a.h;",
    "	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;\n	int trn = 0;\n	TR(0);\n	const int w = a.w, h = a.h;")
rep("		VP2_COMMIT_ALL();\n		if(loader) misc[3] = (int)(pending - a.ticket_base);\n		lds_barrier();",
    "		TR(1);\n		if(tid == 0 && a.trace && trn < 16) a.trace[((size_t)blockIdx.x * 16 + trn) * 8 + 7] = tcur;\n		VP2_COMMIT_ALL();\n		if(loader) misc[3] = (int)(pending - a.ticket_base);\n		lds_barrier();\n		TR(2);")
rep("		lds_barrier();\n\n		// ---- C/D: aggregates, publication, look-back", "		lds_barrier();\n		TR(3);\n\n		// ---- C/D: aggregates, publication, look-back")
rep("		lds_barrier();\n		const int K = misc[1];", "		lds_barrier();\n		TR(4);\n		const int K = misc[1];")
rep("		if(!more) break;\n		rem = next;\n		lds_barrier(); // T16 / Rc / Av / Bv / misc of this tile are dead only now (phase E read them)",
    "		TR(5);\n		trn++;\n		if(!more) break;\n		rem = next;\n		lds_barrier(); // T16 / Rc / Av / Bv / misc of this tile are dead only now (phase E read them)")
rep("		// ---- E: out = local + (sub-segment totals above) + A(x) + B(y) + K, for the lane's own 16 pixels --------------------\n",
    "		TR(6);\n		// ---- E: out = local + (sub-segment totals above) + A(x) + B(y) + K, for the lane's own 16 pixels --------------------\n")
rep("int vp_check_timeout(VpContext* ctx, int stream);",
    "int vp_check_timeout(VpContext* ctx, int stream);\nstatic unsigned long long* g_trace = nullptr;\n"
    "extern \"C\" int vp_dbg_trace_copy(void* host, size_t bytes) { return g_trace ? (int)hipMemcpy(host, g_trace, bytes, hipMemcpyDeviceToHost) : -1; }\n"
    "extern \"C\" int vp_dbg_trace_clear(size_t bytes) { if(!g_trace) { if(hipMalloc(&g_trace, 64u << 20) != hipSuccess) return -1; } return (int)hipMemset(g_trace, 0, bytes); }")
rep("	a.zero = zero3;", "	a.zero = zero3;\n	a.trace = g_trace;")
open(p, "w").write(head + body)
print("patched", p)
