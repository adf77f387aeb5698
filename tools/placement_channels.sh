#!/bin/bash
# Per-channel view of the zero-fill probe on a fast and on a slow block (round-3 verdict item 6): the L2 -> fabric write counters per TCC instance
# (no _sum), JSON output so that the dimension of every sample is kept.  One rocprofv3 pass per counter; the program after -- is python3 itself.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/placement_channels; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for g in "TCC_EA0_WRREQ" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL" "TCC_EA0_WRREQ_STALL"; do
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc $g --output-format json csv -d $O/$i -o p -- python3 $R/tools/placement_pmc.py > $O/$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/$i.log; }
  tail -1 $O/$i.log | cut -c1-300
  i=$((i+1))
done
python3 $R/tools/placement_channels.py $O > $O/summary.json 2> $O/summary.err; head -c 3000 $O/summary.json; tail -3 $O/summary.err
# keep the merged output small: the raw JSON of pass 0 only, compressed
for d in $O/1 $O/2; do find $d -name "*.json" -size +2M -delete; done
find $O -name "*.json" -size +1M -exec gzip -9 {} \;
