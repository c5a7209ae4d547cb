#!/bin/bash
# usage: bash tools/ab_option.sh <option> [runs] — the bench line's headline figures with MISTARK_OPTIONS=<option>=0 and =1, alternating
opt=$1
for i in $(seq ${2:-2}); do for g in 0 1; do
MISTARK_OPTIONS=$opt=$g python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$opt=$g', round(d['value'],1), round(d['ms_per_linear_solve'],4), d.get('linear_solves'), d.get('cg_iterations'), d['host_timers_s']['linear_solve'])"
done; done
