set -u
export TMPDIR=/tmp
bash tools/gpu_round.sh r04b 2>&1 | tail -70
bash tools/r4_adaptive_round.sh r04b_adaptive 2>&1 | tail -12
