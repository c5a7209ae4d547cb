#!/bin/bash
# Collects the per-round profile summaries on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01_v7
# 1. bench.py JSON line, 2. rocprofv3 --kernel-trace --stats of the same command, 3. HBM counters (FETCH_SIZE / WRITE_SIZE) in their own
# --pmc passes (never combined with tracing, as the pool requires). Summaries are written to gpurun_out/<tag>_*.txt; copy them to profiles/.
tag=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
rm -rf /tmp/prof_kt
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python bench.py --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python profiles/summarize_rocpd.py $db > gpurun_out/${tag}_kernel_stats.txt
python profiles/iter_rocpd.py $db > gpurun_out/${tag}_iter.txt
{ python profiles/solve_rocpd.py $db; python profiles/idle_rocpd.py $db 8 0.5; } > gpurun_out/${tag}_timeline.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc
  timeout 900 rocprofv3 --pmc $ctr -d /tmp/prof_pmc -o r -- python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 4 > /tmp/pmc.log 2>&1
  db=$(find /tmp/prof_pmc -name "*.db" | head -1)
  { python profiles/pmc_rocpd.py $db | head -14; python profiles/pmc_rocpd.py $db k_spmv_fused --real; } > gpurun_out/${tag}_pmc_$(echo $ctr | tr A-Z a-z | sed 's/_size//').txt
done
# the bench line LAST, with this build's kernel statistics and counters already in profiles/ (bench.py takes roofline.frac from the newest
# committed kernel trace of its own command and roofline.traffic from the newest counter passes)
cp gpurun_out/${tag}_kernel_stats.txt gpurun_out/${tag}_pmc_fetch.txt gpurun_out/${tag}_pmc_write.txt profiles/
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 400 gpurun_out/${tag}_bench.err
ls -la gpurun_out/${tag}_*
