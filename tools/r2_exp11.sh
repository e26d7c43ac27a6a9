#!/bin/bash
cd $GRAFT_REPO_ROOT
one() { python bench.py --steps 3 --warmup 1 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f us/step'%d['us_per_ssfm_step'], {k: round(v,2) for k,v in r['launch_us'].items()}, 'C1 %.1f us' % d['secondary']['us_per_ssfm_step'])"; }
for r in 1 2; do
for v in product ntstore pnt nocap ldsdb; do L=build/var/_ssfm_$v.so; [ $v = product ] && L=opticomlib_amd/_ssfm_amd.so; echo -n "$v: "; SSFM_LIB=$L one; done
echo -n "lanes=1: "; SSFM_LANES=1 one
echo -n "stagger: "; SSFM_STAGGER=1 one
echo -n "E=8 (k_time): "; SSFM_E=8 SSFM_EF=16 one
echo -n "graph: "; SSFM_GRAPH=1 one
done 2>&1 | tee gpurun_out/r2_knobs.txt
python tools/small_n.py 2>&1 | tail -25 | tee gpurun_out/r2_small_n.txt
