#!/bin/bash
# usage: bash tools/shard_prof.sh W [steps]  — configs[3] on W in-process ranks under rocprofv3: what each rank holds and the per-rank stage times
W=${1:-8}; steps=${2:-8}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/prof_sh
timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/prof_sh -o r -- python tools/shard_model.py $W $steps > gpurun_out/shard${W}_ranks.txt 2>/tmp/sh.err
tail -3 gpurun_out/shard${W}_ranks.txt
db=$(find /tmp/prof_sh -name "*.db" | head -1)
tot=$(grep '^TOTALS' gpurun_out/shard${W}_ranks.txt)
python profiles/shard_rocpd.py $db $W $(echo $tot | cut -d' ' -f2) $(echo $tot | cut -d' ' -f3) > gpurun_out/shard${W}_stages.txt
cat gpurun_out/shard${W}_stages.txt
