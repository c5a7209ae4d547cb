#!/bin/bash
# Round-5 validation on the GPU box (run through gpurun from the repo root): the GPU suite, smoke(), the round's profiles (bench line with all its
# legs; rocprofv3 kernel trace; HBM counters), the HBM-resident SpMV's own kernel trace and counters (88 x 88 x 86 hexahedra = 7.99 M tets), the
# N-process runs of bench.py on the one device (self-launched: pre-flight, RCCL leg refused, stage table), the other configs' rates.
set -x
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-r05_final}
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh $tag 2>&1 | tail -3
# HBM-resident SpMV: kernel trace, then FETCH_SIZE / WRITE_SIZE in their own passes
rm -rf /tmp/p8; GRID=88,88,86 VARIANTS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p8 -o r -- python tools/spmv_sweep.py 2048 > gpurun_out/r05_8M_sweep.txt 2>/dev/null
db=$(find /tmp/p8 -name "*.db" | head -1); python profiles/summarize_rocpd.py $db | head -12 > gpurun_out/r05_8M_kernel_stats.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p8c; GRID=88,88,86 VARIANTS=0 timeout 600 rocprofv3 --pmc $ctr -d /tmp/p8c -o r -- python tools/spmv_sweep.py 2048 > /dev/null 2>&1
  db=$(find /tmp/p8c -name "*.db" | head -1); python profiles/pmc_rocpd.py $db k_spmv_fused > gpurun_out/r05_8M_pmc_$(echo $ctr | tr A-Z a-z | sed 's/_size//').txt
done
for n in 2 4 8; do
  MISTARK_BENCH_DEVICE=0 timeout 900 python bench.py --gpus $n --no-cpu-baseline --no-extras > gpurun_out/r05_shard${n}_bench.json 2> gpurun_out/r05_shard${n}_bench.err
done
MISTARK_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 2 --no-cpu-baseline > gpurun_out/r05_shard2_secondary_bench.json 2> /dev/null
python tools/config_rates.py cfg0 cfg1 cfg2 cfg4 2>&1 | grep "^{" > gpurun_out/r05_config_rates.jsonl
python tools/steplog_cfg2.py 60 2>&1 | grep "^{" >> gpurun_out/r05_config_rates.jsonl
ls -la gpurun_out/r05_* gpurun_out/${tag}_*
