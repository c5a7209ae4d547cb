#!/bin/bash
# Round-4 validation on the GPU box (run through gpurun from the repo root): the GPU suite, smoke(), the round's profiles (bench line with the drop-in,
# secondary and pinned-placement legs; rocprofv3 kernel trace; HBM counters), the N-process runs of bench.py on the one device (self-launched: no
# launcher in the command), the other configs' rates.
set -x
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh ${1:-r04_final} 2>&1 | tail -3
for n in 2 4 8; do
  MISTARK_BENCH_DEVICE=0 timeout 900 python bench.py --gpus $n --no-cpu-baseline --no-extras > gpurun_out/r04_shard${n}_bench.json 2> gpurun_out/r04_shard${n}_bench.err
done
MISTARK_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 2 --no-cpu-baseline > gpurun_out/r04_shard2_secondary_bench.json 2> /dev/null
python tools/config_rates.py cfg0 cfg1 cfg2 cfg4 2>&1 | grep "^{" > gpurun_out/r04_config_rates.jsonl
ls -la gpurun_out/r04_*
