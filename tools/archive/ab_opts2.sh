#!/bin/bash
# usage: bash tools/ab_opts2.sh <steps> "<opts A>" "<opts B>" ...  — A/B of engine options (MISTARK_BENCH_OPTS) on BOTH placements of configs[3], two rounds
steps=$1; shift
for round in 1 2; do
  for o in "$@"; do
    for off in 0,0 0.00137,-0.00053; do
      MISTARK_BENCH_OPTS="$o" timeout 600 python bench.py --no-cpu-baseline --no-extras --steps $steps --offset=$off 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('%-36s %-18s %.2f steps/s  %.3f ms/step  %.3f ms/solve %d solves %d cg' % ('$o' or '(default)', '$off', d['value'], d['ms_per_step'], d['ms_per_linear_solve'], d['linear_solves'], d['cg_iterations']), {k: round(v, 4) for k, v in d['host_timers_s'].items()})"
    done
  done
done
