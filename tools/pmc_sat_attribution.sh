#!/bin/bash
# Attribution of the SAT pass's read traffic, on the GPU box:  bash tools/pmc_sat_attribution.sh
# FETCH_SIZE / WRITE_SIZE of the pass alone (tools/stagebench.py sat 256: flat and the strips prepared once, then the pass over and over) for the shipped
# kernel, for the two result-neutral strip settings and for the ablation builds of tools/abbuild.sh (-DVP_SAT_ABLATE=1|2|3: halo rows / halo columns / both
# not loaded -- wrong tables by design), plus the counter's calibration per access class (tools/fetchcal).  One rocprofv3 --pmc pass per counter and variant.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/satpmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=${B:-256}
run() { # tag, library, extra stagebench args
  tag=$1; lib=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    VP_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$tag/$c -o p -- python3 $R/tools/stagebench.py sat $B --iters 3 "$@" > $O/$tag.$c.log 2>&1 || { echo "$tag $c failed"; tail -3 $O/$tag.$c.log; }
  done
  echo "== $tag"; python3 $R/tools/pmc_sum.py $O/$tag k_sat_fast2
}
run shipped $R/vision-processor_amd/libvp_hip.so
run strips_off $R/vision-processor_amd/libvp_hip.so --no-strips
run abl_rows $R/tools/ab/libvp_hip_abl_rows.so
run abl_cols $R/tools/ab/libvp_hip_abl_cols.so
run abl_both $R/tools/ab/libvp_hip_abl_both.so
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/cal/$c -o p -- $R/tools/fetchcal 2048 > $O/cal.$c.log 2>&1
done
echo "== calibration"; cat $O/cal.FETCH_SIZE.log | grep -E "^k_"; python3 $R/tools/pmc_sum.py $O/cal
