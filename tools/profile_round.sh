#!/bin/bash
# Collects every number DESIGN.md / profiles/ quote, on the GPU box:  bash tools/profile_round.sh  (outputs under gpurun_out/prof/)
# rocprofv3: kernel trace + stats in one run, PMC counters in separate passes (never combined with other trace domains); the program
# after `--` is always python3 itself.
set -e
# PART=1 (default, headline: bench line, batch sweep, kernel stats, HBM bytes), PART=2 (variants, SQ counters, other shapes): each fits one gpurun call
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
PART=${PART:-1}
[ "$PART" = 1 ] && rm -rf $O
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$PART" = 1 ]; then
timeout -k 10 400 python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "bench default done"
for b in 1 8 32 64 128 256 512; do
  timeout -k 10 200 python3 $R/bench.py --batch $b --no-cpu-baseline --no-extras --no-pmc 2>/dev/null | tail -1 > $O/bench_b$b.json
done
echo "batch sweep done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-pmc --prewarm-ms 0 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
echo "kernel stats done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --no-pmc --prewarm-ms 0 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --no-pmc --prewarm-ms 0 --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/pmc_write.log 2>&1
echo "pmc done"
exit 0
fi
# the optional variant without a table (k_flat_walk) and the NEAREST reading of resampling.cl:50 (k_resample_staged<0, 2>): kernel stats + HBM bytes
for v in "local:--sat local" "nearest:--sampler nearest"; do
  tag=${v%%:*}; flags=${v#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o s -- python3 $R/bench.py --no-pmc --prewarm-ms 0 $flags --no-cpu-baseline --no-extras --no-verify > $O/stats_$tag.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$tag -o p -- python3 $R/bench.py --no-pmc --prewarm-ms 0 $flags --no-cpu-baseline --no-extras --no-verify --steps 5 --warmup 2 > $O/pmc_fetch_$tag.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$tag -o p -- python3 $R/bench.py --no-pmc --prewarm-ms 0 $flags --no-cpu-baseline --no-extras --no-verify --steps 5 --warmup 2 > $O/pmc_write_$tag.log 2>&1
done
echo "variants done"
# SQ counters, one pass per group (never combined with other trace domains), default path and the table-free variant
G1="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES"
G2="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G3="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
i=0
for g in "$G1" "$G2" "$G3"; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/sq_default/$i -o p -- python3 $R/bench.py --no-pmc --prewarm-ms 0 --no-cpu-baseline --no-extras --no-verify --steps 5 --warmup 2 > $O/sq_default_$i.log 2>&1
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/sq_local/$i -o p -- python3 $R/bench.py --no-pmc --prewarm-ms 0 --sat local --no-cpu-baseline --no-extras --no-verify --steps 5 --warmup 2 > $O/sq_local_$i.log 2>&1
  i=$((i+1))
done
echo "sq done"
timeout -k 10 200 python3 $R/bench.py --sat twopass --no-pmc --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_twopass.json
for v in tiled twopass; do
  timeout -k 10 200 python3 $R/tools/stagebench.py sat 16 --cfg cfg5_4096x3072_rggb --sat $v 2>/dev/null | grep algorithmic >> $O/stage_cfg5.txt
  timeout -k 10 200 python3 $R/tools/stagebench.py sat 128 --cfg cfg1_640x480_bgr --sat $v 2>/dev/null | grep algorithmic >> $O/stage_cfg1.txt
done
timeout -k 10 200 python3 $R/tools/stagebench.py frame 32 --cfg cfg5_4096x3072_rggb 2>/dev/null | grep algorithmic >> $O/stage_cfg5.txt
timeout -k 10 200 python3 $R/tools/stagebench.py frame 128 --cfg cfg1_640x480_bgr 2>/dev/null | grep algorithmic >> $O/stage_cfg1.txt
echo "all done"
