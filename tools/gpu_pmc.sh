#!/bin/bash
# PMC passes (own runs, no tracing domains mixed in), per the guide.
set -u
TAG=${1:-pmc}
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${TAG}_$C
  timeout 400 rocprofv3 --pmc $C --output-format csv -d gpurun_out/${TAG}_$C -- python3 bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-profile-pass > gpurun_out/${TAG}_$C.json 2> gpurun_out/${TAG}_$C.err; echo "$C rc=$?"
done
python tools/pmc_summary.py gpurun_out/${TAG}_FETCH_SIZE gpurun_out/${TAG}_WRITE_SIZE gpurun_out/${TAG}_traffic.json | tail -40
find gpurun_out/${TAG}_FETCH_SIZE gpurun_out/${TAG}_WRITE_SIZE -name "*.csv" -size +1M -delete
