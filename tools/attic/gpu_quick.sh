#!/bin/bash
# quick GPU check: parity tests + bench (no rocprof)
TAG=${1:-q}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${TAG}_pytest.log
for i in 1 2; do python bench.py --steps 5 --warmup 1 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step  %.2f G/s'%(d['us_per_ssfm_step'], d['value']/1e9), d['roofline']['launch_us'])"; done
