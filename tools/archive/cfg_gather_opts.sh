for o in "" "no_split_gather=1" "no_sym_gather=1" "no_split_gather=1,no_sym_gather=1"; do
for cfg in cfg4 cfg0; do
MISTARK_OPTIONS="$o" python tools/config_rates.py $cfg 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg [$o]', 'steps/s', round(d['newton_steps_per_s'],1), 'newton', d['newton_iterations'], 'solves', d['linear_solves'], 'cg/solve', d['cg_iterations_per_solve'], 'ms/solve', d['ms_per_linear_solve'], d['contact'])"
done; done
