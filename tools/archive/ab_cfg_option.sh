#!/bin/bash
# usage: bash tools/ab_cfg_option.sh <option> cfg0 cfg1 ... — tools/config_rates.py figures with MISTARK_OPTIONS=<option>=0 and =1
opt=$1; shift
for cfg in "$@"; do for g in 0 1; do
MISTARK_OPTIONS=$opt=$g python tools/config_rates.py $cfg 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $opt=$g', 'steps/s', round(d['newton_steps_per_s'],1), 'wall_s', d['wall_s'], 'newton', d['newton_iterations'], 'solves', d['linear_solves'], 'cg/solve', d['cg_iterations_per_solve'], 'ms/solve', d['ms_per_linear_solve'])"
done; done
