export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
for ct in 4 8; do
timeout 300 tools/micro/sym_spmv.bin 56 56 55 $ct 300
done
for ct in 4 8; do
timeout 600 tools/micro/sym_spmv.bin 112 112 110 $ct 50
done
