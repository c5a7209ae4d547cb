#!/bin/bash
# usage: bash tools/ab_variants.sh <variant .so> ... — configs[2] and configs[3] rates with the default build and with each variant library
# (build/variants/*.so, built by hand with another compile-time constant) swapped in for stark_amd/libmistark.so on the GPU box.
cp stark_amd/libmistark.so /tmp/libmistark_default.so
run() {
  python tools/config_rates.py cfg2 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  cfg2 step_ms', round(d['ms_per_newton']['step']/d['time_steps'],3), 'callback_ms', round(d['ms_per_newton']['callback']/d['time_steps'],3))"
  python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  cfg3', round(d['value'],1), 'callback_s', d['host_timers_s']['callback'])"
}
echo default; run
for v in "$@"; do echo $v; cp $v stark_amd/libmistark.so; run; done
cp /tmp/libmistark_default.so stark_amd/libmistark.so
