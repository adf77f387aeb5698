"""Plain device copy rates for reference (tools only; GPU): torch copy_ of the benchmark's flat-sized buffer, several sizes."""
import torch, time
for mb in (100, 400, 1610, 3221):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, dtype=torch.int32, device="cuda").random_(); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"copy {mb} MB: {ms*1e3:.1f} us, {2*n*4/ms/1e6:.0f} GB/s read+write", flush=True)
    w = torch.empty(n, dtype=torch.int32, device="cuda")
    e0.record()
    for _ in range(10): w.fill_(7)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"fill {mb} MB: {ms*1e3:.1f} us, {n*4/ms/1e6:.0f} GB/s write", flush=True)
    e0.record()
    for _ in range(10): s = a.sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"sum  {mb} MB: {ms*1e3:.1f} us, {n*4/ms/1e6:.0f} GB/s read", flush=True)
