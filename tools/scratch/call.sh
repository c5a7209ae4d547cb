export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_contact.py -m gpu -q 2>&1 | tail -12
