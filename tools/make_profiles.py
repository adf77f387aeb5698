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
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
shutil.copy(os.path.join(P, "stats", "s_kernel_stats.csv"), os.path.join(OUT, f"{tag}_kernel_stats.csv"))
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
    for r in csv.DictReader(open(os.path.join(P, name, "p_counter_collection.csv"))):
        per[(r["Kernel_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    acc = collections.defaultdict(list)
    for (k, _), v in per.items():
        acc[k].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


fetch, write = pmc("pmc_fetch"), pmc("pmc_write")
kernels = {}
for k in fetch:
    if not any(t in k for t in ("k_resample_staged", "k_sat_fast", "k_circle_walk", "k_emit")):
        continue
    f_kb, w_kb = fetch[k], write.get(k, 0.0)
    kernels[short(k)] = {"FETCH_SIZE_KB_per_launch": round(f_kb, 1), "WRITE_SIZE_KB_per_launch": round(w_kb, 1),
                         "hbm_bytes_per_launch_corrected": int(2 * f_kb * 1024 + w_kb * 1024)}
old_path = os.path.join(OUT, "r01_pmc_hbm.json")
old = json.load(open(old_path)) if os.path.exists(old_path) else {}
frames = bench["config"]["frames_per_step_per_gpu"]
doc = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2  (tools/profile_round.sh)",
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
json.dump({key: {"hbm_bytes_per_launch": sat["hbm_bytes_per_launch_corrected"], "source": f"profiles/{tag}_pmc_hbm.json", "kernels_hash": bench_mod.kernels_hash()}},
          open(os.path.join(OUT, "hbm_traffic.json"), "w"), indent=1)
sweep = {}
for b in (1, 8, 32, 64, 128, 256):
    d = json.loads(open(os.path.join(P, f"bench_b{b}.json")).read())
    sweep[str(b)] = {"frames_per_s": d["value"], "ms_per_step": d["ms_per_step"], "sat_frac_of_8TBs": d["roofline"]["frac"], "stage_ms": d["stage_ms_per_step"]}
tw = json.loads(open(os.path.join(P, "bench_twopass.json")).read())
extra = {"batch_sweep": sweep,
         "twopass_at_128": {"frames_per_s": tw["value"], "sat_launch_ms": tw["roofline"]["launch_ms"], "sat_frac": tw["roofline"]["frac"]},
         "stagebench_cfg5": open(os.path.join(P, "stage_cfg5.txt")).read().strip().splitlines(),
         "stagebench_cfg1": open(os.path.join(P, "stage_cfg1.txt")).read().strip().splitlines(),
         "extras_of_the_bench_line": bench.get("extras")}
json.dump(extra, open(os.path.join(OUT, f"{tag}_sweeps.json"), "w"), indent=1)
print(json.dumps(doc["kernels"], indent=1))
print(doc["reading"])
