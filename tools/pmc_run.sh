cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /root/repo/gpurun_out/pmc_$tag -o p -- python3 /root/repo/bench.py --batch 128 --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc_$tag.log 2>&1 || exit 1
done
