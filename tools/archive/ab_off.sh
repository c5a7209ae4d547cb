#!/bin/bash
# usage: bash tools/ab_off.sh <steps> "<opts A>" "<opts B>" ...  — like tools/ab.sh on the pinned (offset) placement
steps=$1; shift
for round in 1 2; do
  for o in "$@"; do
    MISTARK_BENCH_OPTS="$o" timeout 600 python bench.py --no-cpu-baseline --no-extras --offset 0.00137,-0.00053 --steps $steps 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('%-32s %.2f steps/s  %.3f ms/step  %.3f ms/solve' % ('$o' or '(default)', d['value'], d['ms_per_step'], d['ms_per_linear_solve']), {k: round(v, 4) for k, v in d['host_timers_s'].items()})"
  done
done
