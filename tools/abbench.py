"""Interleaved A/B timing of one stage between two builds of libvp_hip.so in ONE process on ONE set of buffers (cdna guide rule 24; tools only).
usage: abbench.py STAGE BATCH LIB_A LIB_B     STAGE = resample | sat | tail | local (flat -> blob list, no table) | frame"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load()
stage, B = sys.argv[1], int(sys.argv[2])
libs = sys.argv[3:]

def capi_for(path, tag):
    os.environ["VP_HIP_LIB"] = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location(f"vision_processor_amd.capi_{tag}", os.path.join(ROOT, "vision-processor_amd", "capi.py"), submodule_search_locations=None)
    m = importlib.util.module_from_spec(spec); m.__package__ = "vision_processor_amd"; spec.loader.exec_module(m)
    return m

cfg = vp.configs.ALL[os.environ.get("VP_CFG", "cfg2_2048x1536_rggb")]
mm = cfg.max_matches
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
raw = raw4.repeat((B + 3) // 4, *([1] * (raw4.dim() - 1)))[:B].contiguous()
d = dict(flat=torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
         matches=torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"), idx=torch.zeros((B, mm), dtype=torch.int32, device="cuda"), counter=torch.zeros((B, 3), dtype=torch.int32, device="cuda"))
d_map = torch.zeros((cfg.fh, cfg.fw, 2), dtype=torch.float32, device="cuda")
d_strips = None  # the halo strips vp_frame's resampling writes for the SAT pass: the "sat" stage runs the pass as vp_frame runs it
runs = []
for i, path in enumerate(libs):
    capi = capi_for(path, i)
    ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
    p = capi.frame_params(cfg)
    for kv in filter(None, os.environ.get("VP_TUNE", "").split(",")):  # e.g. VP_TUNE=tail_emit_kernel=1 (applied to every build that knows the knob)
        try: ctx.tune(kv.split("=")[0], int(kv.split("=")[1]))
        except Exception as e: print(f"({os.path.basename(path)}: {e})")
    ctx.frame(p, raw, B, **d); ctx.sync()
    ctx.resample_map(cfg.model(), cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh, d_map)
    if d_strips is None and ctx.halo_strip_bytes(cfg.fw, cfg.fh, cfg.grad_offset):
        d_strips = torch.zeros(ctx.halo_strip_bytes(cfg.fw, cfg.fh, cfg.grad_offset) * B, dtype=torch.uint8, device="cuda")
    if d_strips is not None:
        ctx.halo_strips(d["flat"], cfg.fw, cfg.fh, cfg.grad_offset, B, d_strips); ctx.sync()
    def fn(ctx=ctx, capi=capi, p=p):
        if stage == "resample": ctx.resample_raw_mapped(raw, cfg.layout, cfg.qw, cfg.qh, d_map, cfg.sampler_mode, cfg.fw, cfg.fh, B, d["flat"])
        elif stage == "sat" and d_strips is not None: ctx.graddot_sat_strips(d["flat"], d_strips, cfg.fw, cfg.fh, cfg.grad_offset, B, capi.SAT_TILED, d["sat"])
        elif stage == "sat": ctx.graddot_sat(d["flat"], cfg.fw, cfg.fh, cfg.grad_offset, B, capi.SAT_TILED, d["sat"])
        elif stage == "local": ctx.flat_bloblist(d["flat"], cfg.fw, cfg.fh, B, cfg.grad_offset, cfg.circle_radius, cfg.circ_threshold, cfg.min_score, cfg.list_radius, mm, capi.ORDER_RASTER, d["matches"], d["idx"], d["counter"])
        elif stage == "tail": ctx.blobcenter_list(d["flat"], d["sat"], cfg.fw, cfg.fh, B, cfg.circle_radius, cfg.circ_threshold, cfg.min_score, cfg.list_radius, mm, capi.ORDER_RASTER, d["matches"], d["idx"], d["counter"])
        else: ctx.frame(p, raw, B, **d)
    runs.append((path, fn, ctx))
out = []
for path, fn, ctx in runs:  # outputs must agree bit for bit
    fn(); ctx.sync(); out.append({k: v.clone() for k, v in d.items()})
for k in d:
    if k in ("matches", "idx") or os.environ.get("VP_AB_ABLATION"): continue  # VP_AB_ABLATION=1: the builds are ablations, results differ by design
    assert all(torch.equal(out[0][k], o[k]) for o in out[1:]), f"{k} differs between the builds"
res = [[] for _ in runs]
for rnd in range(8):
    for i, (path, fn, ctx) in enumerate(runs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        if rnd: res[i].append(e0.elapsed_time(e1) * 100)
for (path, _, _), ts in zip(runs, res):
    print(f"{stage} B={B} {os.path.basename(path)}: median {np.median(ts):.1f} us/launch (min {min(ts):.1f})", flush=True)
