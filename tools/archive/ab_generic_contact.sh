for g in 0 1 0 1; do
MISTARK_OPTIONS=generic_contact=$g python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_linear_solve'], d.get('linear_solves'), d.get('cg_iterations'), d.get('newton_iterations'), d['host_timers_s'])"
done
