# second counter set for the resampling kernel: cycles per instruction class, instruction fetch, LDS queue levels (tools/dbg/sq_counters.sh has the first)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sq6
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="GRBM_GUI_ACTIVE SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU"
G2="GRBM_GUI_ACTIVE SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"
G3="GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"
G4="GRBM_GUI_ACTIVE SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL SQ_LEVEL_WAVES"
G5="GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES"
i=0
for g in "$G1" "$G2" "$G3" "$G4" "$G5"; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/$i -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-verify --steps 5 --warmup 2 > $O/log_$i.txt 2>&1 || echo "group $i failed"
  i=$((i+1))
done
echo done
