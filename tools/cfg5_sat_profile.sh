#!/bin/bash
# BASELINE config 5: 4096x3072, LDS-tiled single-pass SAT vs the two-pass ("global-recursive") variant, rocprofv3 kernel times and
# HBM bytes (FETCH_SIZE / WRITE_SIZE in separate passes).  Outputs under gpurun_out/cfg5/.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/cfg5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VP_CFG=cfg5_4096x3072_rggb VP_TWOPASS=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/tools/satbench.py 16 0 > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/tools/satbench.py 16 0 > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/tools/satbench.py 16 0 > $O/write.log 2>&1
grep -h "GB/s algorithmic" $O/stats.log
echo done
