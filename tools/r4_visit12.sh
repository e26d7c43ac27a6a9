set -u
export TMPDIR=/tmp
SSFM_FUSED_PATIENCE_TICKS=-1 python -m pytest tests -m gpu -q -x 2>&1 | tail -12
python -m pytest tests -m gpu -q -x -k "nothing_to_propagate or c_abi" 2>&1 | tail -3
for spec in "14:1" "14:2" "15:1" "15:2" "16:1" "16:2" "17:1" "17:2"; do L=${spec%%:*}; P=${spec#*:}; LOG2N=$L POL=$P python tools/step_time.py fixed_2e${L}x$P; done
python tools/medium_adaptive.py 2>&1 | tail -12
