#!/bin/bash
# usage: bash tools/ab_lib2.sh <steps> libA.so libB.so ...  — A/B of library BUILDS inside one GPU call on BOTH placements of configs[3] (centred: the
# driver's headline; offset: the placement pinned to the reference's log), two rounds each
steps=$1; shift
cp stark_amd/libmistark.so /tmp/libmistark_keep.so
for round in 1 2; do
  for lib in "$@"; do
    cp "$lib" stark_amd/libmistark.so
    for off in 0,0 0.00137,-0.00053; do
      timeout 600 python bench.py --no-cpu-baseline --no-extras --steps $steps --offset=$off 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('%-28s %-18s %.2f steps/s  %.3f ms/step  %.3f ms/solve %d solves' % ('$lib', '$off', d['value'], d['ms_per_step'], d['ms_per_linear_solve'], d['linear_solves']), {k: round(v, 4) for k, v in d['host_timers_s'].items()}, d['contact']['n_detections'])"
    done
  done
done
cp /tmp/libmistark_keep.so stark_amd/libmistark.so
