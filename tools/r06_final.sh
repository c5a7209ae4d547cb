#!/bin/bash
# Round-6 validation on the GPU box (run through gpurun from the repo root): the GPU suite, smoke(), the round's profiles (bench line with all its
# legs incl. value_windows and the N = 1 stage column; rocprofv3 kernel trace; HBM counters), the HBM-resident SpMV's own kernel trace, the
# N-process runs of bench.py on the one device (pre-flight, RCCL leg refused, stage table), the other configs' rates with per-iteration costs.
set -x
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-r06_final}
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh $tag 2>&1 | tail -3
rm -rf /tmp/p8; GRID=88,88,86 VARIANTS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p8 -o r -- python tools/spmv_sweep.py 2048 > gpurun_out/r06_8M_sweep.txt 2>/dev/null
db=$(find /tmp/p8 -name "*.db" | head -1); python profiles/summarize_rocpd.py $db | head -12 > gpurun_out/r06_8M_kernel_stats.txt
for n in 2 8; do
  MISTARK_BENCH_DEVICE=0 timeout 900 python bench.py --gpus $n --no-cpu-baseline --no-extras > gpurun_out/r06_shard${n}_bench.json 2> gpurun_out/r06_shard${n}_bench.err
done
python tools/config_rates.py cfg0 cfg1 cfg2 cfg2tilt cfg4 2>&1 | grep "^{" > gpurun_out/r06_config_rates.jsonl
rm -rf /tmp/pc2; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pc2 -o r -- python tools/config_rates.py cfg2tilt > /dev/null 2>&1
db=$(find /tmp/pc2 -name "*.db" | head -1); python profiles/summarize_rocpd.py $db | head -40 > gpurun_out/r06_cfg2tilt_kernel_stats.txt
ls -la gpurun_out/r06_* gpurun_out/${tag}_*
