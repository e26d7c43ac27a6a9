#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
SSFM_E=16 python -m pytest tests -m gpu -x -q -k "fft or golden" 2>&1 | tail -1
for r in 1 2; do for E in 8 16; do
  echo -n "E=$E single: "; SSFM_E=$E python bench.py --steps 4 --warmup 1 --cpu-steps 0 --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step'%d['us_per_ssfm_step'])"
done; done
for E in 8 16; do echo "E=$E:"; SSFM_E=$E python tools/batch_sweep.py 2>/dev/null | grep -E "fields=(1|4) lanes=2|fields=8 lanes=4"; done
