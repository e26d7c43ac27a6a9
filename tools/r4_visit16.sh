set -u
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -12
timeout 1200 python tests/diag/fuzz_many.py 400 2026 2>&1 | tail -8
