"""Summarise rocprofv3 --pmc counter_collection CSVs under gpurun_out/pmc_*: per kernel, mean counter value per launch."""
import csv, glob, collections, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/pmc_*/p_counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"][:48], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
for k, cs in acc.items():
    if not any(t in k for t in ("k_resample", "k_sat_fast", "k_circle_detect")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
