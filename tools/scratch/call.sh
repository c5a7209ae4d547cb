export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 900 python tools/steplog_cfg2.py 30 0.002 - 3
