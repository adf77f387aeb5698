#!/bin/bash
# Address-translation counters of the zero-fill probe on a fast and on a slow block (tools/placement_pmc.py: 6 sweeps fast, 6 slow, 6 fast), one rocprofv3 pass per group.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/placement_tlb; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for g in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/$i -o p -- python3 $R/tools/placement_pmc.py > $O/$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/$i.log; }
  tail -1 $O/$i.log | cut -c1-250
  i=$((i+1))
done
python3 - $O <<'PY'
import csv, glob, os, sys, collections
for d in sorted(glob.glob(os.path.join(sys.argv[1], "[0-9]"))):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "k_probe_write" not in r["Kernel_Name"]: continue
        per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = per.get(int(r["Dispatch_Id"]), {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(per)[-18:]
    for name in sorted({c for v in per.values() for c in v}):
        vals = [per[i].get(name, 0.0) for i in ids]
        fast = (sum(vals[:6]) + sum(vals[12:])) / 12; slow = sum(vals[6:12]) / 6
        print(f"{name:40s} fast {fast:16.0f}  slow {slow:16.0f}  slow/fast {slow / max(fast, 1):.3f}")
PY
