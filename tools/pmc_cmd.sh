#!/bin/bash
# usage: tools/pmc_cmd.sh OUTDIR "GROUP1 counters" "GROUP2 counters" ... -- python3 script args
# One rocprofv3 --pmc pass per counter group (own run each, with --kernel-trace only), CSV output under gpurun_out/OUTDIR/<n>/.
# The program after -- must be the program itself (python3 ...), never a wrapper.
out=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/gpurun_out/$out"
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${groups[@]}"; do
  timeout -k 10 100 rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$root/gpurun_out/$out/$i" -o p -- "$@" > "$root/gpurun_out/$out/$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$root/gpurun_out/$out/$i.log"; exit 1; }
  i=$((i+1))
done
python3 "$root/tools/pmc_sum.py" "$root/gpurun_out/$out"
