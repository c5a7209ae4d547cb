for i in 1 2 3; do for g in 0 1; do
MISTARK_OPTIONS=no_eval_prelaunch=$g python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['host_timers_s']; print('off' if $g else 'on ', round(d['value'],1), 'eval_pgh', t['eval_pgh'], 'callback', t['callback'], 'newton', t['newton'])"
done; done
