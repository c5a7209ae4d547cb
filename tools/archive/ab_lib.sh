#!/bin/bash
# usage: bash tools/ab_lib.sh <steps> libA.so libB.so ...  — A/B of library BUILDS inside one GPU call (box-to-box spread is 5-8 %): each build is copied over
# stark_amd/libmistark.so in turn, two rounds; prints tools/tet_sweep.py's lazy line and bench.py's rate
steps=$1; shift
cp stark_amd/libmistark.so /tmp/libmistark_keep.so
for round in 1 2; do
  for lib in "$@"; do
    cp "$lib" stark_amd/libmistark.so
    echo "== $lib (round $round)"
    timeout 300 python tools/tet_sweep.py 2>/dev/null | grep "lazy=1 kernel_dbg=[03]"
    timeout 600 python bench.py --no-cpu-baseline --no-extras --steps $steps 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('%.2f steps/s  %.3f ms/step  %.3f ms/solve' % (d['value'], d['ms_per_step'], d['ms_per_linear_solve']), {k: round(v, 4) for k, v in d['host_timers_s'].items()})"
  done
done
cp /tmp/libmistark_keep.so stark_amd/libmistark.so
