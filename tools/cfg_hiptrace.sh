#!/bin/bash
# usage: bash tools/cfg_hiptrace.sh cfg2 — HIP API statistics (host side) of tools/config_rates.py <cfg> -> gpurun_out/<cfg>_hip_api_stats.txt
cfg=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/prof_hip
timeout 900 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/prof_hip -o r -- python tools/config_rates.py $cfg > /tmp/cfg.log 2>&1
grep "^{" /tmp/cfg.log | cut -c1-600
f=$(find /tmp/prof_hip -name "*hip_api_stats.csv" | head -1)
head -25 $f > gpurun_out/${cfg}_hip_api_stats.txt
cat gpurun_out/${cfg}_hip_api_stats.txt
