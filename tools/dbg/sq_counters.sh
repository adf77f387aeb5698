set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sq5
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES"
G2="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G3="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
G4="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
i=0
for g in "$G1" "$G2" "$G3" "$G4"; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $O/$i -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-verify --steps 5 --warmup 2 > $O/log_$i.txt 2>&1 || echo "group $i failed"
  i=$((i+1))
done
echo done
