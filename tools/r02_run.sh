#!/bin/bash
# usage: bash tools/r02_run.sh <tag> [pytest args...]   — GPU tests (optional), bench line, kernel trace + one-iteration dispatch list
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "$1" ]; then timeout 1500 python -m pytest "$@" -x -q 2>&1 | tail -15 > gpurun_out/${tag}_pytest.txt; cat gpurun_out/${tag}_pytest.txt; fi
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 300 gpurun_out/${tag}_bench.err
rm -rf /tmp/prof_kt
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python bench.py --no-cpu-baseline > /tmp/kt.log 2>&1
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $db > gpurun_out/${tag}_kernel_stats.txt
python profiles/iter_rocpd.py $db > gpurun_out/${tag}_iter.txt
{ python profiles/solve_rocpd.py $db; python profiles/idle_rocpd.py $db 8 0.5; } > gpurun_out/${tag}_timeline.txt
python -c "
import json;d=json.load(open('gpurun_out/${tag}_bench.json'));print('BENCH',d['value'],d['ms_per_step'],d['ms_per_linear_solve'],d['cg_iterations_per_solve'],d['host_timers_s'],d['roofline']['frac'])"
