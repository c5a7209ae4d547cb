#!/bin/bash
# usage: bash tools/cfg_abs.sh <cfg> "<kernel name part that opens a unit>" [which]  — absolute-time listing of one unit (time step) of tools/config_rates.py <cfg>
cfg=$1; export ABS_MARK="$2"; w=${3:-5}
export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf /tmp/prof_cfg
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg -o r -- python tools/config_rates.py $cfg > /tmp/cfg.log 2>&1
grep "^{" /tmp/cfg.log | cut -c1-300
db=$(find /tmp/prof_cfg -name "*.db" | head -1)
python profiles/abs_rocpd.py $db $w > gpurun_out/${cfg}_abs.txt
python profiles/summarize_rocpd.py $db > gpurun_out/${cfg}_kernel_stats.txt
