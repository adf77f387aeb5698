#!/bin/bash
# Collects every number DESIGN.md / profiles/ quote, on the GPU box:  bash tools/profile_round.sh  (outputs under gpurun_out/prof/)
# rocprofv3: kernel trace + stats in one run, PMC counters in separate passes (never combined with other trace domains); the program
# after `--` is always python3 itself.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench default done"
for b in 1 8 32 64 128 256; do
  timeout -k 10 200 python3 $R/bench.py --batch $b --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_b$b.json
done
echo "batch sweep done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/stats.log 2>&1
echo "kernel stats done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/pmc_write.log 2>&1
echo "pmc done"
timeout -k 10 200 python3 $R/bench.py --sat twopass --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_twopass.json
for v in tiled twopass; do
  timeout -k 10 200 python3 $R/tools/stagebench.py sat 16 --cfg cfg5_4096x3072_rggb --sat $v 2>/dev/null | grep algorithmic >> $O/stage_cfg5.txt
  timeout -k 10 200 python3 $R/tools/stagebench.py sat 128 --cfg cfg1_640x480_bgr --sat $v 2>/dev/null | grep algorithmic >> $O/stage_cfg1.txt
done
timeout -k 10 200 python3 $R/tools/stagebench.py frame 32 --cfg cfg5_4096x3072_rggb 2>/dev/null | grep algorithmic >> $O/stage_cfg5.txt
timeout -k 10 200 python3 $R/tools/stagebench.py frame 128 --cfg cfg1_640x480_bgr 2>/dev/null | grep algorithmic >> $O/stage_cfg1.txt
echo "all done"
