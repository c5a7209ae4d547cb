export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
bash tools/r06_final.sh r06_final > gpurun_out/r06_final.log 2>&1
grep -E "passed|failed|smoke" gpurun_out/r06_final.log | head
