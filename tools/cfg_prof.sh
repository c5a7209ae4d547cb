#!/bin/bash
# usage: bash tools/cfg_prof.sh cfg2|cfg4|...  — kernel statistics of tools/config_rates.py <cfg> under rocprofv3 -> gpurun_out/<cfg>_kernel_stats.txt
cfg=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/prof_cfg
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg -o r -- python tools/config_rates.py $cfg > /tmp/cfg.log 2>&1
grep "^{" /tmp/cfg.log | cut -c1-400
db=$(find /tmp/prof_cfg -name "*.db" | head -1)
python profiles/summarize_rocpd.py $db > gpurun_out/${cfg}_kernel_stats.txt
head -14 gpurun_out/${cfg}_kernel_stats.txt | cut -c1-150
