#!/bin/bash
# per-dispatch fabric-side counters of the zero-fill probe on a fast and on a slow block (tools/placement_pmc.py), one rocprofv3 pass per group
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/placement_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for g in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
         "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum"; do
  timeout -k 10 60 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/$i -o p -- python3 $R/tools/placement_pmc.py > $O/$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/$i.log; }
  tail -1 $O/$i.log | cut -c1-400
  i=$((i+1))
done
