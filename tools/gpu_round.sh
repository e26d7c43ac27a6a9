#!/bin/bash
# One GPU-box visit that produces everything a round commits under profiles/: parity suite, smoke, the bench lines (C2; C3 and C4 through
# bench.py's own one-rank launch, so that the RCCL gather really runs), rocprofv3 kernel stats of the bench command, PMC traffic,
# SQ counters, and kernel stats of C1 / an adaptive run / the filters.   bash tools/gpu_round.sh TAG   -> gpurun_out/TAG_*
set -u
TAG=${1:-r}
mkdir -p gpurun_out
export TMPDIR=/tmp
T=gpurun_out/$TAG
python -m pytest tests -m gpu -q > ${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 ${T}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > ${T}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 ${T}_smoke.log
python bench.py --steps 5 --warmup 1 > ${T}_bench.json 2> ${T}_bench.err; echo "bench rc=$?"; cut -c1-1800 ${T}_bench.json; tail -2 ${T}_bench.err
python bench.py --workload c3 --steps 2 --warmup 1 --cpu-steps 0 --no-profile-pass > ${T}_bench_c3_1rank.json 2> ${T}_bench_c3.err; echo "bench c3 (one rank, nccl) rc=$?"; cut -c1-900 ${T}_bench_c3_1rank.json
python bench.py --workload c4 --steps 2 --warmup 1 --cpu-steps 0 > ${T}_bench_c4_1rank.json 2> ${T}_bench_c4.err; echo "bench c4 (one rank, nccl) rc=$?"; cut -c1-900 ${T}_bench_c4_1rank.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-steps 0 --no-profile-pass --no-secondary > ${T}_bench_c2_torchrun.json 2> ${T}_bench_c2_torchrun.err; echo "bench under torchrun (one rank) rc=$?"; cut -c1-600 ${T}_bench_c2_torchrun.json
# several ranks on the ONE GPU of this box (diagnostics: BENCH_SAME_GPU + gloo; RCCL wants a GPU per rank): the N > 1 code path of bench.py end to end, and that
# stdout is the JSON line alone
for spec in "2:c2" "2:c4" "4:c4"; do
  N=${spec%%:*}; W=${spec#*:}
  BENCH_SAME_GPU=1 BENCH_DIST_BACKEND=gloo python bench.py --gpus $N --workload $W --steps 1 --warmup 1 --cpu-steps 0 --no-profile-pass --no-secondary > ${T}_bench_${N}ranks_same_gpu_gloo_$W.json 2> ${T}_bench_${N}ranks_$W.err
  echo "bench --gpus $N --workload $W on one GPU (gloo) rc=$? lines=$(wc -l < ${T}_bench_${N}ranks_same_gpu_gloo_$W.json)"; python -c "import json,sys; d=json.load(open('${T}_bench_${N}ranks_same_gpu_gloo_$W.json')); print(d['n_gpus'], d['value'], d['cpu_affinity'])"
done
# where the time of a workgroup goes (in-kernel clock stamps of k_freq alone / with two rows), the lanes' launch timeline and the per-workgroup view (trace build)
{ echo "== k_freq phases, in-kernel clock stamps (tools/stamp_harness.hip), one row per launch"; ./build/stamp_harness 1; echo "== two rows per launch"; ./build/stamp_harness 2;
  echo "== launch timeline of the two lanes (tools/trace_timeline.py, -DSSFM_TRACE=1 build)"; SSFM_LIB=build/var/_ssfm_trace.so python tools/trace_timeline.py 2>&1 | tail -14;
  echo "== per-workgroup view of eight consecutive launches (tools/trace_wg.py)"; SSFM_LIB=build/var/_ssfm_trace.so python tools/trace_wg.py 2>&1 | tail -56; } > ${T}_c2_stamps_and_trace.txt 2>&1; tail -3 ${T}_c2_stamps_and_trace.txt
rm -rf ${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_prof -- python3 bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-profile-pass > ${T}_prof_bench.json 2> ${T}_prof.err; echo "rocprof rc=$?"
find ${T}_prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} ${T}_kernel_stats.csv; head -8 ${T}_kernel_stats.csv
find ${T}_prof -name "*kernel_trace.csv" -size +2M -delete
bash tools/gpu_pmc.sh ${TAG}_pmc > ${T}_pmc.log 2>&1; tail -3 ${T}_pmc.log
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf ${T}_sq
  rocprofv3 --pmc $set --output-format csv -d ${T}_sq -- python3 bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-profile-pass --no-secondary > /dev/null 2> ${T}_sq.err
  python tools/sq_summary.py ${T}_sq "k_freq_c64=k_freq<float" "k_time_mid_c64=k_time<float, 256, 16, 16, 1" > ${T}_sq.txt; cat ${T}_sq.txt
done
find ${T}_sq -name "*.csv" -size +1M -delete
for what in "c1:tools/c1_prof.py" "adaptive:tools/adaptive_prof.py" "sos:tools/sos_prof.py"; do
  name=${what%%:*}; script=${what#*:}
  rm -rf ${T}_${name}prof
  rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_${name}prof -- python3 $script > ${T}_${name}_run.txt 2> ${T}_${name}prof.err
  find ${T}_${name}prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} ${T}_${name}_kernel_stats.csv; head -5 ${T}_${name}_kernel_stats.csv | cut -c1-200
  find ${T}_${name}prof -name "*kernel_trace.csv" -size +1M -delete
done
python tools/cfg_times.py > ${T}_cfg_times.txt 2>&1; cat ${T}_cfg_times.txt
# round 5: SQ counters of the complex128 kernels (configuration C1), its kernels alone / beside each other / with two fields resident, the adaptive long run,
# the capture beside the run, the lanes test 200 times in one process
rm -rf ${T}_c1sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d ${T}_c1sq -- python3 tools/c1_prof.py > /dev/null 2> ${T}_c1sq.err
python tools/sq_summary.py ${T}_c1sq "k_freq_c128=k_freq<double" "k_time_mid_c128=k_time<double, 256, 8, 8, 1" > ${T}_c1_sq.txt; cat ${T}_c1_sq.txt
find ${T}_c1sq -name "*.csv" -size +1M -delete
{ PREC=c128 POL=1 STEPS=100 python tools/step_time.py "C1, one polarisation (one lane: its kernels alone on the chip)";
  PREC=c128 POL=2 STEPS=100 python tools/step_time.py "C1 (two lanes)";
  PREC=c128 POL=2 FIELDS=2 STEPS=100 python tools/step_time.py "C1, two fields resident (four lanes)";
  python tools/attic/adapt_long.py; } > ${T}_c1_alone.txt 2>&1; cat ${T}_c1_alone.txt
python tools/capture_time.py > /dev/null 2>&1; cp gpurun_out/r05_capture_time.txt ${T}_capture_time.txt; cat ${T}_capture_time.txt
python tests/diag/lane_stability.py 200 > /dev/null 2>&1; cp gpurun_out/r05_lane_stability.txt ${T}_lane_stability.txt; tail -1 ${T}_lane_stability.txt
# round 5, late: the one-launch filter after the hand-over change (shapes, the phase timeline of every workgroup -- needs build/var/_ssfm_tl.so -- SQ counters
# of k_filtfilt), and the cost of a blocked stream-wait packet (tools/attic/barrier_cost.hip, built into build/barrier_cost)
{ for r in 1 2 3; do echo "== round $r"; python3 tools/filter_shapes.py; done; } > ${T}_sos_shapes.txt 2>&1; tail -6 ${T}_sos_shapes.txt
if [ -f build/var/_ssfm_tl.so ]; then
  { echo "== 2^20 x 2 complex128"; SSFM_LIB=$PWD/build/var/_ssfm_tl.so SOS_TL_PATH=$PWD/${T}_sos_timeline_all.txt python3 tools/sos_timeline.py
    echo "== 2^16 real"; SSFM_LIB=$PWD/build/var/_ssfm_tl.so LOG2N=16 ROWS=1 CPLX=0 python3 tools/sos_timeline.py; } > ${T}_sos_timeline.txt 2>&1
fi
[ -x build/barrier_cost ] && { timeout 60 ./build/barrier_cost 5; timeout 60 ./build/barrier_cost 10; timeout 90 ./build/barrier_cost 40; } > ${T}_barrier_cost.txt 2>&1
bash tools/gpu_pmc_sos.sh ${TAG}_sos_pmc > ${T}_sos_sq.txt 2>&1; tail -6 ${T}_sos_sq.txt
# the randomised capture stress (two seeds) and the fixed costs of an adaptive run
{ python3 tests/diag/capture_stress.py 120 5 | tail -3; python3 tests/diag/capture_stress.py 120 11 | tail -3; } > ${T}_capture_stress.txt 2>&1; tail -2 ${T}_capture_stress.txt
python3 tools/attic/adaptive_fixed_costs.py > ${T}_adaptive_fixed_costs.txt 2>&1; cat ${T}_adaptive_fixed_costs.txt
