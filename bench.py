#!/usr/bin/env python3
"""Benchmark of the blob-detection hot path on MI355X (contract: see the task's bench.py section; BASELINE.json metric).

One "step" = one pass of the whole hot path (raw2quad+resampling -> gradientDot+SAT -> circle+blobList) over one
batch of `--batch` synthetic 2048x1536 RGGB frames already resident in HBM.  value = frames/s over all ranks.
Ranks shard independent camera feeds (no data-path collective); with N > 1 the per-step detection records are
all-gathered over RCCL (the only exchange the path has).

`python bench.py --gpus N` with no RANK in the environment starts its N rank processes itself (fresh children, spawned
before this process has touched a GPU or imported torch) and relays rank 0's JSON line; under torchrun (RANK set) it is one rank.

Extra objects on the JSON line:
  roofline     -- the gradientDot+SAT pass: algorithmic 8 B/px (read flat once, write SAT once) x pixels per launch
                  / mean stage duration from hipEvents recorded on the launch stream during the timed region.
  cpu_baseline -- the CPU checkers timed on a bounded sample of the same workload (rank 0, N=1); see cpu_baseline().
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def _cgroup_cpu_quota():
    """Smallest CPU quota (in CPUs) along this process's cgroup path, v2 (cpu.max) or v1 (cfs quota / period); None if unlimited."""
    best = None
    try:
        lines = open("/proc/self/cgroup").read().strip().splitlines()
    except Exception:
        lines = []
    rels = {ln.split(":", 2)[2] for ln in lines if ln.count(":") >= 2} | {"/"}
    for rel in rels:
        for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"):
            p = rel
            while True:
                d = os.path.join(base, p.lstrip("/"))
                try:
                    txt = open(os.path.join(d, "cpu.max")).read().split()
                    if txt[0] != "max":
                        q = int(txt[0]) / int(txt[1])
                        best = q if best is None else min(best, q)
                except Exception:
                    pass
                try:
                    q = int(open(os.path.join(d, "cpu.cfs_quota_us")).read())
                    if q > 0:
                        q = q / int(open(os.path.join(d, "cpu.cfs_period_us")).read())
                        best = q if best is None else min(best, q)
                except Exception:
                    pass
                if p in ("", "/"):
                    break
                p = os.path.dirname(p)
    return best


def host_cores():
    """CPU share of this process from what the kernel says, nothing guessed: min(affinity mask, cgroup CPU quota along the cgroup
    path).  VP_BENCH_THREADS overrides (recorded in the bench line as `cores`, the threads actually used)."""
    if os.environ.get("VP_BENCH_THREADS"):
        return max(1, int(os.environ["VP_BENCH_THREADS"]))
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    q = _cgroup_cpu_quota()
    if q is not None:
        n = min(n, max(1, int(q + 0.5)))
    return n


def opencl_devices():
    """Every OpenCL platform and device of this box with type, CL_DEVICE_IMAGE_SUPPORT and version: whether ANY real OpenCL runtime
    here could run the reference's unmodified image kernels (resampling.cl:50 needs images + samplers; setup.sh:27-29 makes PoCL the
    reference's CPU runtime).  Pure enumeration through the ICD loader; run in a child process before any HIP call (bench line)."""
    import ctypes as C
    try:
        cl = C.CDLL("libOpenCL.so.1")
    except OSError:
        return {"loader": None, "platforms": [], "summary": "no OpenCL ICD loader on this box"}
    n = C.c_uint(0)
    if cl.clGetPlatformIDs(0, None, C.byref(n)) != 0 or n.value == 0:
        return {"loader": "libOpenCL.so.1", "platforms": [], "summary": "OpenCL loader present, 0 platforms"}
    plats = (C.c_void_p * n.value)()
    cl.clGetPlatformIDs(n, plats, None)

    def pinfo(p, what):
        buf = C.create_string_buffer(512)
        return buf.value.decode(errors="replace") if cl.clGetPlatformInfo(C.c_void_p(p), C.c_uint(what), C.c_size_t(512), buf, None) == 0 else None

    def dstr(d, what):
        buf = C.create_string_buffer(512)
        return buf.value.decode(errors="replace") if cl.clGetDeviceInfo(C.c_void_p(d), C.c_uint(what), C.c_size_t(512), buf, None) == 0 else None

    types = {1: "DEFAULT", 2: "CPU", 4: "GPU", 8: "ACCELERATOR", 16: "CUSTOM"}
    out, n_cpu, n_img = [], 0, 0
    for p in plats:
        entry = {"name": pinfo(p, 0x0902), "version": pinfo(p, 0x0901), "devices": []}
        m = C.c_uint(0)
        rc = cl.clGetDeviceIDs(C.c_void_p(p), C.c_ulong(0xFFFFFFFF), 0, None, C.byref(m))
        if rc == 0 and m.value:
            devs = (C.c_void_p * m.value)()
            cl.clGetDeviceIDs(C.c_void_p(p), C.c_ulong(0xFFFFFFFF), m, devs, None)
            for d in devs:
                t = C.c_ulong(0)
                cl.clGetDeviceInfo(C.c_void_p(d), C.c_uint(0x1000), C.c_size_t(8), C.byref(t), None)
                img = C.c_uint(0)
                cl.clGetDeviceInfo(C.c_void_p(d), C.c_uint(0x1016), C.c_size_t(4), C.byref(img), None)
                tn = "|".join(v for k, v in types.items() if t.value & k) or hex(t.value)
                entry["devices"].append({"name": dstr(d, 0x102B), "type": tn, "image_support": bool(img.value), "device_version": dstr(d, 0x102F),
                                         "opencl_c_version": dstr(d, 0x103D)})
                n_cpu += bool(t.value & 2)
                n_img += bool(img.value)
        else:
            entry["clGetDeviceIDs"] = rc
        out.append(entry)
    ndev = sum(len(e["devices"]) for e in out)
    summary = f"{len(out)} OpenCL platform(s), {ndev} device(s), {n_cpu} CPU device(s), {n_img} with image support"
    if n_img == 0:
        summary += ": no runtime on this box can run the reference's image kernels (resampling.cl:50), so nothing here can pin the oracle"
    return {"loader": "libOpenCL.so.1", "platforms": out, "summary": summary}


def opencl_devices_child():
    """opencl_devices() in a child process: the AMD OpenCL runtime brings its own ROCr instance, which must not meet PyTorch's HIP runtime."""
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--opencl-probe"], capture_output=True, text=True, timeout=120)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"summary": f"OpenCL probe failed: {e!r}"}


def kernels_hash():
    """Identity of the kernel sources a traffic figure was measured on (profiles/hbm_traffic.json carries the same hash)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vision-processor_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_live(args, kernel_substr, seconds=90):
    """HBM bytes per launch of the roofline kernel, measured IN THIS RUN: two child passes of this very command under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`
    / `--pmc WRITE_SIZE` (counters in their own passes, nothing else traced; the program after `--` is python3 itself), each a short run of the same workload, batch
    and SAT variant.  FETCH_SIZE is doubled -- on gfx950 it tallies every 128-byte line at 64 B, for every load width (MI355X_MICROARCH.md, HBM section; calibrated per
    access class in profiles/r06_pmc_sat_attribution.json) -- WRITE_SIZE is exact; both are KB.  Returns (bytes, description) or (None, reason): never raises, the headline does not
    depend on it.  The parent keeps its buffers while the children run (they fit beside them many times over)."""
    import glob, shutil, tempfile, csv
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as d:
                cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                       "--batch", str(args.batch), "--workload", args.workload, "--sat", args.sat, "--order", args.order, "--sampler", args.sampler, "--steps", "2", "--warmup", "1",
                       "--prewarm-ms", "0", "--no-cpu-baseline", "--no-extras", "--no-verify"]
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=seconds, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
                tot, n = 0.0, 0
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if kernel_substr in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                            tot += float(row["Counter_Value"])
                            n += 1
                if n == 0:
                    return None, f"no {counter} rows for {kernel_substr} (rocprofv3 rc {r.returncode}: {(r.stderr or r.stdout)[-200:]!r})"
                vals[counter] = tot / n
    except Exception as e:  # a profiler that is missing, refuses or hangs must not take the bench line with it
        return None, f"live PMC passes failed: {e!r}"
    total = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return total, (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command inside this run (2 steps each), mean per dispatch of {kernel_substr}: "
                   f"reads {2.0 * vals['FETCH_SIZE'] * 1024 / 1e6:.1f} MB (FETCH_SIZE x 2, gfx950), writes {vals['WRITE_SIZE'] * 1024 / 1e6:.1f} MB")


def extra_measurements(capi, torch, np, cfg, params, d_raw, flat, sat, matches, counter, idx, B):
    """Measurements beside the headline (BASELINE.json configs[2] and the figures DESIGN.md quotes), taken AFTER the timed region on the same
    resident frames; none of them is `value`.  Returns a dict for the `extras` key."""
    out = {}
    mm = cfg.max_matches
    ctx = capi.Context(0, n_streams=8)  # library-owned streams: frames of different cameras in flight
    try:
        def rate(fn, n, frames_per_call):
            for _ in range(3):
                fn()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            ctx.sync()
            return frames_per_call * n / (time.perf_counter() - t0)

        # frames per launch group (one stream): throughput against batching
        sweep = {}
        for b in sorted({1, 8, 32, 128, B}):
            if b > B:
                continue
            sweep[str(b)] = round(rate(lambda: ctx.frame(params, d_raw, b, flat, sat, matches, counter, idx=idx, stream=0), max(4, min(200, 1024 // b)), b), 1)
        out["frames_per_s_by_frames_per_launch"] = sweep
        # The optional variant without a global table (sat_variant = VP_SAT_LOCAL: gradientDot + sliding box sums + circle + blobList in one
        # kernel, flat read once, buffers.sat = NULL).  Same blob list bit for bit (tests/test_gpu_round2.py); the headline keeps the SAT pass.
        p_local = capi.frame_params(cfg, sat_variant=capi.SAT_LOCAL, order=params.order)
        local = {}
        for b in sorted({1, 8, 32, 128, B}):
            if b > B:
                continue
            local[str(b)] = round(rate(lambda: ctx.frame(p_local, d_raw, b, flat, None, matches, counter, idx=idx, stream=0), max(4, min(200, 1024 // b)), b), 1)
        out["frames_per_s_sat_local_variant"] = local
        # the other candidate of the unpinned sampler (resampling.cl:50): NEAREST through the same staged kernel family
        if cfg.sampler_mode != 0:
            p_near = capi.frame_params(cfg, sat_variant=params.sat_variant, order=params.order, sampler_mode=0)
            near_rate = round(rate(lambda: ctx.frame(p_near, d_raw, B, flat, sat, matches, counter, idx=idx, stream=0), max(4, min(200, 1024 // B)), B), 1)
            ctx.events_clear()
            ctx.events_enable(True)
            for _ in range(5):
                ctx.frame(p_near, d_raw, B, flat, sat, matches, counter, idx=idx, stream=0)
            ctx.events_enable(False)
            st = {}
            for name, ms in ctx.event_times():
                st.setdefault(name, []).append(ms)
            ctx.events_clear()
            out["frames_per_s_sampler_nearest"] = near_rate
            out["stage_ms_sampler_nearest"] = {k: round(float(np.mean(v)), 4) for k, v in st.items()}
        ctx.frame(params, d_raw, B, flat, sat, matches, counter, idx=idx, stream=0)  # leave the default path's outputs in the buffers
        ctx.sync()
        # one frame, host-synchronised after every frame: latency of the whole hot path
        n = 200
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.frame(params, d_raw, 1, flat, sat, matches, counter, idx=idx, stream=0)
            ctx.sync(0)
        out["single_frame_latency_us"] = round(1e6 * (time.perf_counter() - t0) / n, 1)
        # BASELINE configs[2]: s streams in flight, b frames per launch on each (own buffers per stream)
        per = 8
        bufs = [dict(flat=torch.empty((per, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.empty((per, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
                     m=torch.zeros((per, 22 * mm), dtype=torch.uint8, device="cuda"), i=torch.zeros((per, mm), dtype=torch.int32, device="cuda"),
                     c=torch.zeros((per, 3), dtype=torch.int32, device="cuda")) for _ in range(8)]
        torch.cuda.synchronize()
        streams = {}
        for ns, b in ((1, 1), (2, 1), (4, 1), (8, 1), (2, 8), (4, 8), (8, 8)):
            def go():
                for st in range(ns):
                    u = bufs[st]
                    ctx.frame(params, d_raw[st * b:], b, u["flat"], u["sat"], u["m"], u["c"], idx=u["i"], stream=st)
            streams[f"{ns}x{b}"] = round(rate(go, 100 if b == 1 else 30, ns * b), 1)
        out["frames_per_s_streams_x_frames_per_launch"] = streams
        # PCIe-inclusive: pinned host frames -> H2D -> vp_frame -> D2H of counters + records, 8 frames per copy and launch group, two streams
        g = min(per, B)
        h_raw = torch.from_numpy(np.ascontiguousarray(d_raw[:g].cpu().numpy())).pin_memory()
        h_m = [torch.zeros((g, 22 * mm), dtype=torch.uint8).pin_memory() for _ in range(2)]
        h_c = [torch.zeros((g, 3), dtype=torch.int32).pin_memory() for _ in range(2)]
        d_in = [torch.empty_like(d_raw[:g]) for _ in range(2)]
        torch.cuda.synchronize()
        lib = capi.lib()
        import ctypes as C

        def pcie(k):
            st = k & 1
            u = bufs[st]
            lib.vp_copy_h2d(ctx.h, st, C.c_void_p(d_in[st].data_ptr()), C.c_void_p(h_raw.data_ptr()), C.c_size_t(h_raw.numel()))
            ctx.frame(params, d_in[st], g, u["flat"], u["sat"], u["m"], u["c"], idx=u["i"], stream=st)
            lib.vp_copy_d2h(ctx.h, st, C.c_void_p(h_m[st].data_ptr()), C.c_void_p(u["m"].data_ptr()), C.c_size_t(g * 22 * mm))
            lib.vp_copy_d2h(ctx.h, st, C.c_void_p(h_c[st].data_ptr()), C.c_void_p(u["c"].data_ptr()), C.c_size_t(g * 12))
        kk = [0]

        def pcie_step():
            pcie(kk[0])
            kk[0] += 1
        out["frames_per_s_pcie_inclusive"] = round(rate(pcie_step, 40, g), 1)
        # the same with vp_ctx_tune "frame_graph" = 1: the three launches of a repeated vp_frame call replayed as one captured hipGraph
        ctx.tune("frame_graph", 1)
        for _ in range(3):
            ctx.frame(params, d_raw, 1, flat, sat, matches, counter, idx=idx, stream=0)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.frame(params, d_raw, 1, flat, sat, matches, counter, idx=idx, stream=0)
            ctx.sync(0)
        out["single_frame_latency_us_frame_graph"] = round(1e6 * (time.perf_counter() - t0) / n, 1)
        gstreams = {}
        for ns, b in ((1, 1), (8, 1), (8, 8)):
            def go():
                for st in range(ns):
                    u = bufs[st]
                    ctx.frame(params, d_raw[st * b:], b, u["flat"], u["sat"], u["m"], u["c"], idx=u["i"], stream=st)
            gstreams[f"{ns}x{b}"] = round(rate(go, 100 if b == 1 else 30, ns * b), 1)
        out["frames_per_s_streams_x_frames_per_launch_frame_graph"] = gstreams
        ctx.tune("frame_graph", 0)
        out["cfg5_sat_variants"] = cfg5_sat_variants(capi, torch, np, ctx)
        out["note"] = "measured after the timed region on the same resident frames (PCIe figure: host-pinned frames copied in, records copied out); none of these is `value`"
    finally:
        ctx.close()
    return out


def cfg5_sat_variants(capi, torch, np, ctx, B=32):
    """BASELINE configs[4]: the 4096x3072 feed, LDS-tiled single-pass SAT against the global-recursive (row pass + column pass) variant, as HBM
    bandwidth of the gradientDot+SAT pass on its algorithmic 8 B per flat pixel.  32 frames per launch (4 distinct rendered frames, repeated), buffers from
    vp_frame_buffers_create; hipEvents around the stage on the launch stream; one frame of each variant's last launch held to the CPU oracle."""
    from _vp_pkg import load
    from concurrent.futures import ThreadPoolExecutor
    vp = load()
    cfg = vp.configs.CFG5
    mm = cfg.max_matches
    with ThreadPoolExecutor(4) as pool:
        distinct = list(pool.map(lambda i: cfg.synth(vp.feeds.feed_seed(5, i)), range(4)))
    raws = np.stack([distinct[i % 4] for i in range(B)])
    d_raw = torch.from_numpy(raws).cuda()
    out = {"workload": cfg.name, "frames_per_launch": B, "algorithmic_bytes_per_launch": 8.0 * cfg.n_flat * B, "peak_gbs": HBM_PEAK_GBS}
    for name, sv in (("tiled", capi.SAT_TILED), ("twopass", capi.SAT_TWOPASS)):
        p = capi.frame_params(cfg, sat_variant=sv, order=capi.ORDER_RASTER)
        slab = ctx.frame_slab(p, B)
        try:
            for _ in range(3):
                ctx.frame_buffers(p, d_raw, B, slab.b)
            ctx.sync()
            ctx.events_clear()
            ctx.events_enable(True)
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                ctx.frame_buffers(p, d_raw, B, slab.b)
            ctx.sync()
            dt = time.perf_counter() - t0
            ctx.events_enable(False)
            st = {}
            for tag, ms in ctx.event_times():
                st.setdefault(tag, []).append(ms)
            ctx.events_clear()
            ms = float(np.mean(st["gradientDot+sat"]))
            gbs = 8.0 * cfg.n_flat * B / (ms * 1e-3) / 1e9
            flat, sat, m_t, idx_t, cnt_t = slab.torch_views()
            ver = verify_against_oracle(cfg, raws, [B - 1], flat, sat, m_t, idx_t, cnt_t, mm)
            del flat, sat, m_t, idx_t, cnt_t
            out[name] = {"launch_ms": round(ms, 4), "achieved_gbs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "frames_per_s_whole_path": round(B * n / dt, 1),
                         "stage_ms": {k: round(float(np.mean(v)), 4) for k, v in st.items()}, "verified": ver["ok"], "mismatches": ver["mismatches"],
                         "sat_block_write_gbs": slab.placement["sat"]["write_gbs"]}
        finally:
            slab.destroy()
    return out


def spawn_ranks(n, argv, timeout_s=1500.0):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent has not imported torch nor touched a
    GPU), relay rank 0's JSON line.  All children are polled: the first one that exits non-zero (a bad device, an OOM, an assertion)
    ends the others at once instead of leaving them in a rendezvous or a barrier until the collective's own timeout, and its stderr
    tail is what the caller sees; an overall timeout bounds the whole run.  Only the exact PIDs started here are ever signalled."""
    import socket
    import tempfile
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, logs = [], []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        logs.append(tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=logs[r]))

    def tail(f, lines=40):
        f.seek(0)
        return "\n".join(f.read().decode(errors="replace").splitlines()[-lines:])

    t0, failed = time.time(), None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = (bad[0], f"rank {bad[0]} exited with code {rcs[bad[0]]}")
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.time() - t0 > timeout_s:
            failed = (next(r for r, rc in enumerate(rcs) if rc is None), f"no result after {timeout_s:.0f} s")
            break
        time.sleep(0.2)
    if failed:
        for p in procs:  # the ranks this function started, by handle
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        r, why = failed
        sys.stderr.write(f"bench.py --gpus {n}: {why}; the other ranks were stopped.  stderr of rank {r}:\n{tail(logs[r])}\n")
        return 1
    for r in range(n):  # library chatter of every rank, prefixed
        for line in tail(logs[r], 20).splitlines():
            sys.stderr.write(f"[rank {r}] {line}\n")
    out0.seek(0)
    for line in out0.read().decode().splitlines():  # exactly the JSON line goes to stdout; anything else rank 0 printed goes to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return 0


def verify_against_oracle(cfg, raws, frames, flat, sat, matches, idx, counter, mm):
    """The timed launch checked against the CPU oracle (the checker, never the measured path): the given frames of the LAST timed step --
    flat, the int32 SAT, counters, raster-sorted pixel indices and the 22-byte records -- byte for byte.  Returns the `verified` object of
    the bench line; a mismatch makes bench.py exit non-zero instead of printing a number."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    port = O.Port(threads=host_cores())
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    scratch = port.frame_scratch(cfg.qw, cfg.qh, cfg.fw, cfg.fh, mm)
    bad = []
    for k in frames:
        m, widx, cnt = port.frame(raws[k], cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                                  cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm, scratch=scratch)
        if not np.array_equal(flat[k].cpu().numpy(), scratch["flat"]):
            bad.append((k, "flat"))
        if sat is not None:
            wsat, mx = port.sat_i32(scratch["graddot"])
            if mx >= 2 ** 24 or not np.array_equal(sat[k].cpu().numpy(), wsat):
                bad.append((k, "sat"))
        c = counter[k].cpu().numpy()
        if not np.array_equal(c, cnt):
            bad.append((k, f"counter {c.tolist()} != {cnt.tolist()}"))
            continue
        n = min(int(c[0]), mm)
        if not np.array_equal(idx[k].cpu().numpy().view(np.uint32)[:n], widx):
            bad.append((k, "pixel indices"))
        if matches[k].cpu().numpy()[:22 * n].tobytes() != m.tobytes():
            bad.append((k, "records"))
    return {"frames": [int(k) for k in frames], "ok": not bad, "against": "oracle/vp_oracle.c (CPU checker) on the same sensor frames: flat, int32 SAT, counters, sorted pixel indices, 22-byte records, byte for byte",
            "mismatches": [f"frame {k}: {what}" for k, what in bad]}


def cpu_baseline(cfg, seconds, opencl):
    """The CPU checkers timed on the host cores, never part of the measured GPU path.  kind "port" = this repo's plain-C restatement
    (oracle/vp_oracle.c), the faster of the two and the one no rule questions.  Nested `reference_kernels_with_builtin_shim` (when oracle/_ref
    is present) = the reference's kernel/*.cl compiled unmodified for x86 but linked against oracle/ref_shim.c -- this repo's stand-in for
    the OpenCL builtin library (work-item ids, image reads / writes, samplers), since no OpenCL runtime that could run them exists here (see
    `opencl`): the kernel arithmetic is the reference's compiled code, the image semantics are this repo's reading of the spec, so it is
    NOT "the reference run here" and is reported for information only."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    cores = host_cores()
    port = O.Port(threads=cores)
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    raws = [cfg.synth(1000 + i) for i in range(4)]
    scratch = port.frame_scratch(cfg.qw, cfg.qh, cfg.fw, cfg.fh, cfg.max_matches)

    def one(i):
        port.frame(raws[i % len(raws)], cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                   cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, cfg.max_matches, sat_mode=0, scratch=scratch)

    for i in range(2):
        one(i)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        one(n)
        n += 1
    dt = time.perf_counter() - t0
    out = {"value": round(n / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port", "opencl_runtimes": opencl.get("summary"),
           "sample": f"{n} frames of {cfg.name} through oracle/vp_oracle.c vpo_frame (fp32 sequential SAT, rows sharded over {cores} pthreads), {dt:.1f} s"}
    if O.Ref.available():
        ref = O.Ref()
        ref_scratch = (np.zeros(4 * cfg.qw * cfg.qh, np.uint8), np.zeros(4 * cfg.fw * cfg.fh, np.uint8), np.zeros(cfg.n_flat, np.float32), np.zeros(cfg.n_flat, np.float32),
                       np.zeros(cfg.n_flat, np.float32))

        def ref_one(i, threads):
            return ref.frame_mt(raws[i % len(raws)], cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                                cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, cfg.max_matches, threads, ref_scratch)

        ref_one(0, cores)
        rn, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            ref_one(rn, cores)
            rn += 1
        rdt = time.perf_counter() - t0
        t1 = time.perf_counter()
        ref_one(0, 1)
        single = 1.0 / (time.perf_counter() - t1)
        out["reference_kernels_with_builtin_shim"] = {
            "value": round(rn / rdt, 3), "unit": "frames/s", "cores": cores,
            "sample": f"{rn} frames of {cfg.name} through the reference's kernel/*.cl compiled unmodified for x86 (oracle/_ref) and linked against oracle/ref_shim.c, "
                      f"this repo's stand-in for the OpenCL builtin library (NOT a real OpenCL runtime: none that can run image kernels exists on this box); "
                      f"each NDRange in row bands over {cores} host threads, {rdt:.1f} s; one thread: {single:.3f} frames/s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (one launch group); 0 = feeds.BENCH_FRAMES_PER_STEP (256)")
    ap.add_argument("--workload", default="cfg2_2048x1536_rggb")
    ap.add_argument("--sat", default="tiled", choices=["tiled", "twopass", "local"], help="tiled = the single-pass SAT the roofline is quoted on (default); twopass = global-recursive comparison; "
                    "local = optional variant without a global table (one fused kernel after resampling)")
    ap.add_argument("--order", default="raster", choices=["raster", "atomic"])
    ap.add_argument("--sampler", default="bilinear", choices=["bilinear", "nearest"], help="what read_imageui through CLK_FILTER_LINEAR on an integer image returns (resampling.cl:50; OpenCL leaves "
                    "it undefined, SURVEY H1): bilinear = spec 8.2 formula truncated (default, the headline), nearest = point sampling")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="bounded CPU sample, per baseline (reference kernels and port)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detect", action="store_true", help="BASELINE config 4 in full: additionally run the CPU hypothesis stage (main.cpp:319-370, one detector per feed "
                    "slot, host threads, overlapped with the next step's kernels) and gather the encoded SSL_WrapperPackets instead of the blob records. Off by default: "
                    "the headline metric is the GPU hot path.")
    ap.add_argument("--detect-threads", type=int, default=0, help="host threads of the CPU stage (0 = this rank's share of the cores)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra measurements taken after the headline region (batch sweep, single-frame latency, "
                    "streams in flight, PCIe-inclusive rate); they run at N = 1 only and never enter `value`")
    ap.add_argument("--plain-alloc", action="store_true", help="frame buffers from plain hipMalloc (no placement check): the allocation lottery of DESIGN.md section 4")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of two frames of the last timed step with the CPU oracle (rank 0, after the timed region)")
    ap.add_argument("--opencl-probe", action="store_true", help="print the OpenCL platform / device report as JSON and exit (used as a child process)")
    ap.add_argument("--prewarm-ms", type=float, default=1500.0, help="untimed load before the warm-up steps so that the device's clocks have settled (0 = none)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic live (two short child runs of this command under rocprofv3 --pmc); the figure then comes from "
                    "profiles/hbm_traffic.json if its kernel hash matches, else null")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) this process IS one of those child runs: no children of its own")
    args = ap.parse_args()
    if args.opencl_probe:
        print(json.dumps(opencl_devices()))
        return 0
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])  # nothing GPU-related has been imported or called in this process

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    # the OpenCL report comes from a child process started before this one touches the GPU
    opencl = opencl_devices_child() if rank == 0 and world == 1 and not args.no_cpu_baseline else {}
    import numpy as np
    import torch
    from _vp_pkg import load
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no fallback"
    # VP_BENCH_REHEARSE=1 (tools only): every rank on cuda:0 with the gloo backend, to rehearse the N > 1 code path on a one-GPU box
    rehearse = os.environ.get("VP_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # VP_BENCH_FORCE_DIST=1 (tools only): initialise RCCL and run the per-step gather even at world size 1, so that the collective
    # path (backend "nccl" = RCCL, uint8 all_gather_into_tensor on the launch stream's tensors) can be exercised on a one-GPU box
    force_dist = os.environ.get("VP_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    vp = load()
    capi = vp.load_capi()
    cfg = vp.configs.ALL[args.workload]
    if args.sampler == "nearest":  # the other reading of resampling.cl:50: everything below -- kernels, oracle check, CPU baseline -- follows
        cfg = cfg.with_(sampler_mode=vp.configs.SAMPLER_NEAREST)
    B, mm = args.batch or vp.feeds.BENCH_FRAMES_PER_STEP, cfg.max_matches
    args.batch = B
    ctx = capi.Context(local_rank, stream_ptr=torch.cuda.current_stream().cuda_stream)
    params = capi.frame_params(cfg, sat_variant={"tiled": capi.SAT_TILED, "twopass": capi.SAT_TWOPASS, "local": capi.SAT_LOCAL}[args.sat],
                               order=capi.ORDER_RASTER if args.order == "raster" else capi.ORDER_ATOMIC)

    # synthetic feeds: distinct seeds per rank and frame, uploaded once (inputs resident in HBM before the timed region)
    feeds = vp.feeds
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, min(16, host_cores() // max(1, world)))) as pool:  # the renderer is C++ behind ctypes: the GIL is released
        raws = np.stack(list(pool.map(lambda i: cfg.synth(feeds.feed_seed(rank, i)), range(B))))
    d_raw = torch.from_numpy(raws).cuda()
    # every buffer vp_frame writes comes from the library's allocator, as a deployment's would: allocated once, the table and flat from
    # placement-checked blocks (vp_frame_buffers_create; --plain-alloc: take what hipMalloc gives, the round-2 behaviour)
    if args.plain_alloc:
        ctx.tune("placement_candidates", 0)
    elif not rehearse:
        ctx.tune("placement_max_mib", -1)  # one rank per GPU and nobody else on it: the placement search may hold nine tenths of the free memory (default: a quarter of the device, with a reserve of an eighth)
    slab = ctx.frame_slab(params, B)
    flat, sat, matches, idx, counter = slab.torch_views()
    # what a rank contributes to the per-step gather: the COMPACT records (vp_records_compact: per frame counter[3], n and the n records a host loop
    # reads, main.cpp:299-301) -- 0.85 MB per rank and step instead of the 6.7 MB of the fixed-capacity arrays
    cap = feeds.COMPACT_CAP
    rec_bytes = feeds.compact_block_bytes(B, cap)
    record = [torch.zeros(rec_bytes, dtype=torch.uint8, device="cuda") for _ in range(2)]  # double-buffered: the gather of step k overlaps step k+1
    gathered = [torch.empty((world, rec_bytes), dtype=torch.uint8, device="cuda") for _ in range(2)] if dist else None
    pending = [None, None]
    step_no = [0]

    det = None
    if args.detect:
        # host tensor copies must not wake an OpenMP team sized for every visible CPU (256 on a GPU box with a 16-core quota: the
        # spinning team eats the quota the detector threads need -- measured 40 ms per step instead of 3)
        torch.set_num_threads(1)
        hostlib = vp.hostlib
        dcfg = hostlib.config_defaults()
        det = {"done": 0, "threads": args.detect_threads or (host_cores() if world == 1 else max(1, min(16, len(os.sched_getaffinity(0)) // world))), "detectors": [], "ev": [torch.cuda.Event() for _ in range(2)],
               "m": [torch.zeros_like(matches) for _ in range(2)], "c": [torch.zeros_like(counter) for _ in range(2)],
               "hm": [torch.zeros((B, 22 * mm), dtype=torch.uint8).pin_memory() for _ in range(2)], "hc": [torch.zeros((B, 3), dtype=torch.int32).pin_memory() for _ in range(2)],
               "hp": torch.zeros((B, feeds.PACKET_CAP), dtype=torch.uint8), "hl": torch.zeros((B,), dtype=torch.int32),
               "hrec": [torch.zeros((B, feeds.packet_record_bytes()), dtype=torch.uint8).pin_memory() for _ in range(2)],
               "drec": [torch.zeros((B, feeds.packet_record_bytes()), dtype=torch.uint8, device="cuda") for _ in range(2)],
               "gath": [torch.empty((world, B, feeds.packet_record_bytes()), dtype=torch.uint8, device="cuda") for _ in range(2)] if dist else None}
        for i in range(B):
            dcfg.cam_id = rank * B + i  # every feed slot is its own camera: own colour model, own tracker
            d = hostlib.Detector(dcfg)
            lines, arcs = cfg.field_markings()
            d.set_geometry_direct(hostlib.nadir_calib(cfg.model()), (cfg.qw, cfg.qh), cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh,
                                  field=(int(cfg.field_len), int(cfg.field_wid), 300, -1, 10), lines=lines, arcs=arcs)
            det["detectors"].append(d)

    def cpu_stage(p, frame_no):
        """Hypotheses -> SSL_DetectionFrame -> SSL_WrapperPacket for the B feeds of the step whose read-back sits in host buffer p."""
        det["ev"][p].synchronize()
        t = frame_no / 70.0
        hostlib.detect_batch(det["detectors"], det["hm"][p].numpy(), det["hc"][p].numpy(), mm, frame_no, t, t, t + 0.002, det["threads"], det["hp"].numpy(), det["hl"].numpy())
        if pending[p] is not None:
            pending[p].wait()
            pending[p] = None
        feeds.pack_packets(det["hl"], det["hp"], det["hrec"][p])  # single-threaded host copy (torch.set_num_threads(1) above)
        det["drec"][p].copy_(det["hrec"][p], non_blocking=True)
        if dist:
            pending[p] = feeds.gather_records(det["drec"][p], det["gath"][p], dist, async_op=True)

    def cpu_upto(j):
        while det["done"] < j:
            det["done"] += 1
            cpu_stage((det["done"] - 1) & 1, det["done"])

    def step():
        if det is not None:
            k = step_no[0]
            s = k & 1
            step_no[0] += 1
            ctx.frame(params, d_raw, B, flat, sat, det["m"][s], det["c"][s], idx=idx)
            det["hm"][s].copy_(det["m"][s], non_blocking=True)
            det["hc"][s].copy_(det["c"][s], non_blocking=True)
            det["ev"][s].record()
            cpu_upto(k)  # the previous step's blobs (steps are numbered from 1) while the GPU works on this one
            return
        ctx.frame(params, d_raw, B, flat, sat, matches, counter, idx=idx)
        if dist:
            s = step_no[0] & 1
            step_no[0] += 1
            if pending[s] is not None:
                pending[s].wait()  # the gather issued two steps ago has released record[s] / gathered[s]
            ctx.records_compact(counter, matches, idx, B, mm, cap, record[s])
            pending[s] = feeds.gather_records(record[s], gathered[s], dist, async_op=True)

    def drain():
        if det is not None:
            cpu_upto(step_no[0])  # the last step's CPU stage has no following GPU step to hide behind
        for s in range(2):
            if pending[s] is not None:
                pending[s].wait()
                pending[s] = None

    # Clock warm-up (untimed, before the W warm-up steps): on some boxes the first ~0.3 s of load run 2-3 % slower in EVERY kernel and then speed up in steps,
    # whatever the buffers (profiles/r06_clock_warmup.json: per-stage times of one process over its first second; fresh processes with 3 / 100 / 400 warm-up
    # steps) -- the device's clock governor, not the kernels.  W = 3 steps are 9 ms: without this the timed region of a fresh process sits inside that ramp.
    # A fixed NUMBER of steps, the same on every rank (a step of the multi-rank path issues a collective): --prewarm-ms at the ~11.7 us per frame of this path.
    n_pre = min(20000, int(args.prewarm_ms / max(0.08, 0.0117 * B))) if args.prewarm_ms > 0 else 0
    for i in range(n_pre):
        step()
        if (i & 7) == 7:
            torch.cuda.synchronize()  # keep the queue short: the host runs far ahead of 3 ms steps otherwise
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    ctx.sync()  # raises VP_ERR_TIMEOUT if a bounded look-back wait of the single-pass SAT expired
    # every step runs the same frames: what the timed steps produce must equal what the warm-up produced
    warm_cnt = (det["c"][(step_no[0] - 1) & 1] if det is not None and step_no[0] else counter).cpu().numpy().copy()
    warm_sat = sat[B - 1, cfg.fh - 1, cfg.fw - 1].item() if args.warmup and sat is not None else None
    ctx.events_clear()
    ctx.events_enable(True)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()  # every step's gather has completed inside the timed region
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.events_enable(False)
    ctx.sync()  # a SAT look-back timeout inside the timed region invalidates the run: fail instead of printing a number
    cnt_now = (det["c"][(step_no[0] - 1) & 1] if det is not None else counter).cpu().numpy()
    if args.warmup:
        assert np.array_equal(cnt_now, warm_cnt), "the timed steps' blob counters differ from the warm-up's on identical input"
        assert sat is None or sat[B - 1, cfg.fh - 1, cfg.fw - 1].item() == warm_sat, "the timed steps' SAT differs from the warm-up's on identical input"
    assert int(cnt_now[:, 0].min()) >= cfg.n_robots * 5, f"fewer blobs than rendered robot discs: {cnt_now[:, 0].min()}"
    stage_ms = {}
    for name, ms in ctx.event_times():
        stage_ms.setdefault(name, []).append(ms)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    coll = "gloo (one-GPU rehearsal)" if rehearse else "RCCL"
    failed = False
    if rank == 0:
        cnt = cnt_now
        if args.sat == "local":  # no SAT pass exists: the fused kernel reads flat once (4 B/px algorithmic)
            roof_kernel, sat_ms, algo_bytes = "gradientDot+circle+blobList from sliding box sums, no table (reads flat once)", float(np.mean(stage_ms["gradientDot+satBlobCenter+blobList (local)"])), 4.0 * cfg.n_flat * B
        else:
            roof_kernel, sat_ms, algo_bytes = "gradientDot+SAT pass", float(np.mean(stage_ms["gradientDot+sat"])), 8.0 * cfg.n_flat * B
        achieved = algo_bytes / (sat_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes of this same command (tools/profile_round.sh -> profiles/hbm_traffic.json), valid only
        # for the kernel sources it was measured on: the entry carries their hash, a mismatch gives null
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            ent = json.load(open(tpath)).get(f"{args.workload}/b{B}/{args.sat}", {})
            if ent.get("kernels_hash") == kernels_hash():
                traffic, traffic_src = ent.get("hbm_bytes_per_launch"), ent.get("source")
        under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)  # never a profiler inside a profiler
        if world == 1 and not args.no_pmc and not args.pmc_child and not under_profiler:
            live, why = pmc_traffic_live(args, {"tiled": "k_sat_fast2", "local": "k_flat_walk"}[args.sat]) if args.sat in ("tiled", "local") else (None, "the two-pass variant is several kernels")
            if live is not None:
                traffic, traffic_src = round(live), why
            else:
                traffic_src = (traffic_src or "none") + f" [live PMC unavailable: {why}]"
        out = {
            "metric": "frames/s at 2048x1536 (whole hot path raw2quad->blobList, inputs resident in HBM)",
            "value": round(world * B * args.steps / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (fp32 for coordinates and scores)", "data": "synthetic",
            "config": {"workload": cfg.name, "frames_per_step_per_gpu": B, "sensor": "2048x1536 RGGB8", "flat": f"{cfg.fw}x{cfg.fh}", "grad_offset": cfg.grad_offset,
                       "circle_radius": cfg.circle_radius, "list_radius": cfg.list_radius, "max_blobs": mm, "sat_variant": args.sat, "list_order": args.order, "prewarm_ms": args.prewarm_ms,
                       "sampler": "NEAREST" if args.sampler == "nearest" else "BILINEAR_RTZ", "sharding": f"{world} independent feeds, 1 per GPU" + (f", {coll} all-gather of detection records per step" if dist else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": roof_kernel, "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": round(sat_ms, 4)},
            "stage_ms_per_step": {k: round(float(np.mean(v)), 4) for k, v in stage_ms.items()},
            "blobs_per_frame": float(np.minimum(cnt[:, 0], mm).mean()),
            "placement": dict(slab.placement, note="zero-fill GB/s of the blocks vp_frame_buffers_create kept for the table and flat, of the candidates it probed"),
        }
        for k, pl in slab.placement.items():  # a search that ran out of budget kept a slow block for the TABLE: say so where the number is read
            if k == "sat" and pl.get("warning"):
                out.setdefault("warnings", []).append(f"placement[{k}]: {pl['warning']}")
                print(f"[WARN] bench.py: placement[{k}]: {pl['warning']}", file=sys.stderr)
        # every stage against the same HBM roofline, on its own algorithmic bytes per flat pixel: raw sensor byte + flat
        # written (resampling), flat read + table written (SAT pass), table read (circle + blobList), flat read (local variant)
        per_px = {"raw2quad+resampling": 4.0 + cfg.qw * cfg.qh * (3.0 if cfg.layout == vp.configs.BGR else 4.0) / cfg.n_flat, "gradientDot+sat": 8.0, "satBlobCenter+blobList": 4.0,
                  "gradientDot+satBlobCenter+blobList (local)": 4.0}
        # the whole path on its compulsory bytes: raw sensor in + flat out / in + table out / in (no table with --sat local: raw + flat out / in)
        path_bytes = cfg.raw_bytes + (8.0 if args.sat == "local" else 16.0) * cfg.n_flat
        path_gbs = path_bytes * B * args.steps / elapsed / 1e9  # per GPU: every rank runs B frames per step
        out["path_roofline"] = {"bound": "hbm", "bytes_per_frame": path_bytes, "achieved": round(path_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(path_gbs / HBM_PEAK_GBS, 4),
                                "note": "compulsory bytes of one frame x frames per step / ms_per_step of this rank's GPU (whole hot path, all launches and gaps included)"}
        out["stage_roofline_frac"] = {k: round(per_px[k] * cfg.n_flat * B / (float(np.mean(v)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for k, v in stage_ms.items() if k in per_px}
        if dist and det is None:
            # what the last step gathered: rank 0's own slots must be what vp_frame left in its arrays, and no slot may have been too small
            gc, gn, gm, gi = feeds.unpack_compact(gathered[(step_no[0] - 1) & 1], B, cap)
            assert np.array_equal(gc[rank].numpy(), cnt_now), "gathered counters differ from this rank's own"
            assert np.array_equal(gn.numpy(), np.minimum(gc[:, :, 0].numpy(), mm)), f"a frame has more than {cap} blobs: raise feeds.COMPACT_CAP"
            assert np.array_equal(gi[rank][0].numpy(), idx[0, :int(gn[rank, 0])].cpu().numpy())
            out["gather"] = {"bytes_per_rank_and_step": rec_bytes, "full_record_bytes": B * feeds.record_bytes(mm), "form": f"compact records, {cap} per frame slot"}
        if det is not None:
            # what the last step emitted: parse this rank's packets back (and, for N > 1, check that the gather delivered every rank's)
            torch.cuda.synchronize()
            last = (step_no[0] - 1) & 1
            mine = feeds.unpack_packets(det["drec"][last][None])[0]
            probe = vp.hostlib.Detector()
            robots, balls = [], []
            for pkt in mine:
                has, src = probe.decode(pkt)
                assert has and src == 3
                fr = probe.frame()
                robots.append(len(fr["robots_yellow"]) + len(fr["robots_blue"]))
                balls.append(len(fr["balls"]))
            out["metric"] = "frames/s at 2048x1536 (hot path + CPU hypothesis stage + SSL_WrapperPacket gather)"
            out["detect"] = {"host_threads": det["threads"], "robots_per_frame": float(np.mean(robots)), "balls_per_frame": float(np.mean(balls)),
                             "packet_bytes": float(np.mean([len(p) for p in mine])), "rendered": {"robots": cfg.n_robots, "balls": cfg.n_balls}}
            if dist:
                allp = feeds.unpack_packets(det["gath"][last])
                out["detect"]["gathered_packets"] = sum(1 for r in allp for p in r if p)
                assert allp[0] == mine
            out["config"]["sharding"] = f"{world} x {B} independent feeds" + (f", {coll} all-gather of SSL_WrapperPackets per step" if dist else "")
        if not args.no_verify:
            # what the timed steps left in the buffers (before the extra measurements reuse them), held to the CPU oracle
            vm, vc = (det["m"][(step_no[0] - 1) & 1], det["c"][(step_no[0] - 1) & 1]) if det is not None else (matches, counter)
            out["verified"] = verify_against_oracle(cfg, raws, sorted({0, B // 2, B - 1}), flat, sat, vm, idx, vc, mm)
        if world == 1 and not args.no_extras and det is None and sat is not None:
            out["extras"] = extra_measurements(capi, torch, np, cfg, params, d_raw, flat, sat, matches, counter, idx, B)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds, opencl)
            out["opencl"] = opencl
        if out.get("verified") and not out["verified"]["ok"]:
            sys.stderr.write("bench.py: the timed launch does NOT match the CPU oracle: " + "; ".join(out["verified"]["mismatches"]) + "\n")
            failed = True
        else:
            print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    del flat, sat, matches, idx, counter
    slab.destroy()
    ctx.close()
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
