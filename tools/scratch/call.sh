export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
for o in "" "sweep_axis_by_extent=1" "" "sweep_axis_by_extent=1"; do
MISTARK_OPTIONS="$o" python tools/config_rates.py cfg4 cfg0 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('[$o]', d['config'], d['newton_steps_per_s'], d['newton_iterations'], d['linear_solves'], d['ms_per_newton']['callback'], d['ms_per_newton_iteration'])"
MISTARK_OPTIONS="$o" python bench.py --no-cpu-baseline --no-extras --offset 0.00137,-0.00053 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$o] bench offset', d['value'], d['host_timers_s']['callback'], d['linear_solves'], d['cg_iterations'])"
done
