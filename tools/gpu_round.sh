#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (N = 1: C2; one-rank RCCL run of the C3 path), rocprofv3 kernel trace.
# Outputs under gpurun_out/ (copy what is to be kept into profiles/).
set -u
TAG=${1:-r}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.log
python bench.py --steps 5 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --workload c3 --steps 2 --warmup 1 --cpu-steps 0 --no-profile-pass > gpurun_out/${TAG}_bench_c3_1rank.json 2> gpurun_out/${TAG}_bench_c3.err; echo "bench c3 (1 rank, nccl) rc=$?"; cut -c1-900 gpurun_out/${TAG}_bench_c3_1rank.json; tail -3 gpurun_out/${TAG}_bench_c3.err
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-profile-pass > gpurun_out/${TAG}_prof_bench.json 2> gpurun_out/${TAG}_prof.err; echo "rocprof rc=$?"
find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12
find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/${TAG}_kernel_stats.csv
# keep only the summaries (the full trace is large)
find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -size +2M -delete
