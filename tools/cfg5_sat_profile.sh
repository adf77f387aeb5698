#!/bin/bash
# BASELINE configs[4]: 4096x3072, 32 frames per launch, LDS-tiled single-pass SAT vs the two-pass ("global-recursive") variant, rocprofv3 kernel times and
# HBM bytes (FETCH_SIZE / WRITE_SIZE in separate passes).  Outputs under gpurun_out/cfg5/.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/cfg5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in tiled twopass; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -o s -- python3 $R/tools/stagebench.py sat 32 --cfg cfg5_4096x3072_rggb --sat $v > $O/stats_$v.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -o p -- python3 $R/tools/stagebench.py sat 32 --cfg cfg5_4096x3072_rggb --sat $v --iters 2 > $O/fetch_$v.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$v -o p -- python3 $R/tools/stagebench.py sat 32 --cfg cfg5_4096x3072_rggb --sat $v --iters 2 > $O/write_$v.log 2>&1
  grep -h "GB/s algorithmic" $O/stats_$v.log
done
echo done
