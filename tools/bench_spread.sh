#!/bin/bash
# N fresh bench.py processes back to back on one box: the run-to-run spread of the headline and of the SAT pass.
# usage: tools/bench_spread.sh OUTFILE [N=8] [extra bench.py flags...]
out=$1; n=${2:-8}; shift; shift
: > "$out"
for i in $(seq 1 "$n"); do
	python bench.py --no-pmc --no-extras --no-cpu-baseline --no-verify "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'frames_per_s': d['value'], 'sat_frac': d['roofline']['frac'], 'stage_ms': d['stage_ms_per_step'], 'placement': {k: v for k, v in d.get('placement', {}).items() if k != 'note'}}))" >> "$out"
done
cat "$out"
