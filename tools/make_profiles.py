"""Turns the raw outputs of tools/profile_round.sh (gpurun_out/prof/) into the committed summaries under profiles/."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"


def _csv_under(name, suffix):
    for dirpath, _, files in os.walk(os.path.join(P, name)):
        for fn in files:
            if fn.endswith(suffix):
                return os.path.join(dirpath, fn)
    raise FileNotFoundError(f"{name}: no *{suffix}")


shutil.copy(_csv_under("stats", "kernel_stats.csv"), os.path.join(OUT, f"{tag}_kernel_stats.csv"))
bench = json.loads(open(os.path.join(P, "bench_default.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(OUT, f"{tag}_bench_n1.json"), "w"), indent=1)
# the bench line printed by the very process rocprofv3 traced: its hipEvent figure for the SAT pass is the one that must agree with
# the kernel_stats average (the pass moves by +-7 % from process to process on one box, so two separate runs need not agree)
for line in reversed(open(os.path.join(P, "stats.log")).read().splitlines()):
    if line.startswith("{") and '"roofline"' in line:
        traced = json.loads(line)
        json.dump(traced, open(os.path.join(OUT, f"{tag}_bench_under_rocprof.json"), "w"), indent=1)
        break


def pmc(name):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(_csv_under(name, "counter_collection.csv"))):
        per[(r["Kernel_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    acc = collections.defaultdict(list)
    for (k, _), v in per.items():
        acc[k].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def hbm_rows(fetch, write, into, names):
    for k in fetch:
        if not any(t in k for t in names):
            continue
        f_kb, w_kb = fetch[k], write.get(k, 0.0)
        into[short(k)] = {"FETCH_SIZE_KB_per_launch": round(f_kb, 1), "WRITE_SIZE_KB_per_launch": round(w_kb, 1),
                          "hbm_bytes_per_launch_corrected": int(2 * f_kb * 1024 + w_kb * 1024)}


kernels = {}
hbm_rows(pmc("pmc_fetch"), pmc("pmc_write"), kernels, ("k_resample_staged", "k_sat_fast", "k_circle_walk", "k_emit"))
for variant, names in (("local", ("k_flat_walk",)), ("nearest", ("k_resample_staged",))):  # bench.py --sat local / --sampler nearest
    if os.path.isdir(os.path.join(P, f"pmc_fetch_{variant}")):
        hbm_rows(pmc(f"pmc_fetch_{variant}"), pmc(f"pmc_write_{variant}"), kernels, names)
    st = os.path.join(P, f"stats_{variant}")
    if os.path.isdir(st):
        shutil.copy(_csv_under(f"stats_{variant}", "kernel_stats.csv"), os.path.join(OUT, f"{tag}_kernel_stats_{variant}.csv"))
old = {}
for prev in ("r05_pmc_hbm.json", "r04_pmc_hbm.json", "r03_pmc_hbm.json", "r02_pmc_hbm.json"):  # the correction note and the membench calibration are carried along
    if os.path.exists(os.path.join(OUT, prev)):
        old = json.load(open(os.path.join(OUT, prev)))
        break
frames = bench["config"]["frames_per_step_per_gpu"]
doc = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --no-pmc --no-cpu-baseline --no-extras --steps 5 --warmup 2  (tools/profile_round.sh)",
    "workload": f"{bench['config']['workload']}, {frames} frames per launch",
    "correction": old.get("correction", ""),
    "kernels": kernels,
    "calibration": old.get("calibration", {}),
}
sat = [v for k, v in kernels.items() if "k_sat_fast" in k][0]
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
doc["reading"] = {"k_sat_fast": f"algorithmic {algo/1e6:.1f} MB per launch (8 B/px); measured {sat['hbm_bytes_per_launch_corrected']/1e6:.1f} MB = "
                                f"{sat['hbm_bytes_per_launch_corrected']/algo:.2f}x: writes {sat['WRITE_SIZE_KB_per_launch']*1024/1e6:.0f} MB, "
                                f"reads {2*sat['FETCH_SIZE_KB_per_launch']*1024/1e6:.0f} MB"}
if "history" in old:
    doc["history"] = old["history"]
json.dump(doc, open(os.path.join(OUT, f"{tag}_pmc_hbm.json"), "w"), indent=1)
key = f"{bench['config']['workload']}/b{frames}/tiled"
sys.path.insert(0, ROOT)
import bench as bench_mod  # the hash of the kernel sources the traffic figure belongs to (bench.py emits null on a mismatch)
traffic = {key: {"hbm_bytes_per_launch": sat["hbm_bytes_per_launch_corrected"], "source": f"profiles/{tag}_pmc_hbm.json", "kernels_hash": bench_mod.kernels_hash()}}
fw = [v for k, v in kernels.items() if "k_flat_walk" in k]
if fw:
    traffic[f"{bench['config']['workload']}/b{frames}/local"] = {"hbm_bytes_per_launch": fw[0]["hbm_bytes_per_launch_corrected"], "source": f"profiles/{tag}_pmc_hbm.json", "kernels_hash": bench_mod.kernels_hash()}
json.dump(traffic, open(os.path.join(OUT, "hbm_traffic.json"), "w"), indent=1)
# SQ counters: mean per dispatch and kernel, default path and the table-free variant
sq_lines = ["rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py [--sat local] --no-cpu-baseline --no-extras --no-verify --steps 5 --warmup 2  (tools/profile_round.sh, three passes each; cfg2, " + str(frames) + " frames per launch)",
            "mean counter value per dispatch.  SQ_ACTIVE_INST_VALU is in ~4-cycle units per SIMD: busy fraction = 4 x value / (1024 SIMDs x GRBM_GUI_ACTIVE / 8);",
            "SQ_ACTIVE_INST_LDS in ~2-cycle units per CU: 2 x value / (256 CUs x GRBM_GUI_ACTIVE / 8); SALU: one issue slot per SIMD every 4 cycles.", ""]
for run in ("sq_default", "sq_local"):
    d = os.path.join(P, run)
    if not os.path.isdir(d):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for dirpath, _, files in os.walk(d):
        for fn in files:
            if fn.endswith("counter_collection.csv"):
                per = collections.defaultdict(float)
                for r in csv.DictReader(open(os.path.join(dirpath, fn))):
                    per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
                for (k, _, c), v in per.items():
                    a = acc[short(k)][c]
                    a[0] += v
                    a[1] += 1
    sq_lines.append(f"== {run} ==")
    for k, cs in sorted(acc.items()):
        if not any(t in k for t in ("k_resample_staged", "k_sat_fast", "k_circle_walk", "k_flat_walk", "k_emit_queue")):
            continue
        sq_lines.append(k)
        for c, (sv, n) in sorted(cs.items()):
            sq_lines.append(f"   {c:45s} {sv / n:16.1f}   (x{n})")
open(os.path.join(OUT, f"{tag}_pmc_sq.txt"), "w").write("\n".join(sq_lines) + "\n")
sweep = {}
for b in (1, 8, 32, 64, 128, 256, 512):
    d = json.loads(open(os.path.join(P, f"bench_b{b}.json")).read())
    sweep[str(b)] = {"frames_per_s": d["value"], "ms_per_step": d["ms_per_step"], "sat_frac_of_8TBs": d["roofline"]["frac"], "stage_ms": d["stage_ms_per_step"]}
tw = json.loads(open(os.path.join(P, "bench_twopass.json")).read())
extra = {"batch_sweep": sweep,
         f"twopass_at_{frames}": {"frames_per_s": tw["value"], "sat_launch_ms": tw["roofline"]["launch_ms"], "sat_frac": tw["roofline"]["frac"]},
         "stagebench_cfg5": open(os.path.join(P, "stage_cfg5.txt")).read().strip().splitlines(),
         "stagebench_cfg1": open(os.path.join(P, "stage_cfg1.txt")).read().strip().splitlines(),
         "extras_of_the_bench_line": bench.get("extras")}
json.dump(extra, open(os.path.join(OUT, f"{tag}_sweeps.json"), "w"), indent=1)
print(json.dumps(doc["kernels"], indent=1))
print(doc["reading"])
