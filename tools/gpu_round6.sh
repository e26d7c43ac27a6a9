#!/bin/bash
# Round 6: one GPU-box visit that produces what the round commits under profiles/ (r06_final_*).   bash tools/gpu_round6.sh [TAG]
set -u
TAG=${1:-r06_final}
mkdir -p gpurun_out
export TMPDIR=/tmp
T=gpurun_out/$TAG
rm -f gpurun_out/parity_margins.txt
python -m pytest tests -m gpu -q > ${T}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 ${T}_pytest.log
cp gpurun_out/parity_margins.txt ${T}_parity_margins.txt
python -c "import __graft_entry__ as g; g.smoke()" > ${T}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 ${T}_smoke.log
python bench.py --steps 5 --warmup 1 > ${T}_bench.json 2> ${T}_bench.err; echo "bench rc=$?"; cut -c1-1500 ${T}_bench.json; tail -2 ${T}_bench.err
python bench.py --workload c3 --steps 2 --warmup 1 --cpu-steps 0 --no-profile-pass > ${T}_bench_c3_1rank.json 2> ${T}_bench_c3.err; echo "bench c3 (one rank, nccl) rc=$?"; cut -c1-700 ${T}_bench_c3_1rank.json; grep "bench rank" ${T}_bench_c3.err | tail -4
python bench.py --workload c4 --steps 2 --warmup 1 --cpu-steps 0 > ${T}_bench_c4_1rank.json 2> ${T}_bench_c4.err; echo "bench c4 (one rank, nccl) rc=$?"; cut -c1-700 ${T}_bench_c4_1rank.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-steps 0 --no-profile-pass --no-secondary > ${T}_bench_c2_torchrun.json 2> ${T}_bench_c2_torchrun.err; echo "bench under torchrun (one rank) rc=$?"; cut -c1-400 ${T}_bench_c2_torchrun.json
for spec in "2:c2" "2:c4" "4:c4"; do
  N=${spec%%:*}; W=${spec#*:}
  BENCH_SAME_GPU=1 BENCH_DIST_BACKEND=gloo python bench.py --gpus $N --workload $W --steps 1 --warmup 1 --cpu-steps 0 --no-profile-pass --no-secondary --launch-timeout 300 > ${T}_bench_${N}ranks_same_gpu_gloo_$W.json 2> ${T}_bench_${N}ranks_$W.err
  echo "bench --gpus $N --workload $W on one GPU (gloo) rc=$? lines=$(wc -l < ${T}_bench_${N}ranks_same_gpu_gloo_$W.json)"; python -c "import json,sys; d=json.load(open('${T}_bench_${N}ranks_same_gpu_gloo_$W.json')); print(d['n_gpus'], d['ranks_seen'], d['value'], d['us_per_ssfm_step_by_rank'])"
done
# the launch budget on the real thing: two ranks, one of which never shows up (a rank count the box does not have GPUs for under RCCL) must end within the budget
timeout 200 python bench.py --gpus 2 --steps 1 --warmup 0 --cpu-steps 0 --no-profile-pass --no-secondary --launch-timeout 60 > ${T}_bench_2ranks_nccl_one_gpu.json 2> ${T}_bench_2ranks_nccl_one_gpu.err; echo "bench --gpus 2 over RCCL on a one-GPU box rc=$? (expected: non-zero, within the budget)"; tail -3 ${T}_bench_2ranks_nccl_one_gpu.err | cut -c1-300
{ echo "== launch timeline of the two lanes (tools/trace_timeline.py, -DSSFM_TRACE=1 build)"; SSFM_LIB=build/var/_ssfm_trace.so python tools/trace_timeline.py 2>&1 | tail -14; } > ${T}_c2_trace.txt 2>&1; tail -3 ${T}_c2_trace.txt
rm -rf ${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_prof -- python3 bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-profile-pass --no-big-field > ${T}_prof_bench.json 2> ${T}_prof.err; echo "rocprof rc=$?"
find ${T}_prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} ${T}_kernel_stats.csv; head -8 ${T}_kernel_stats.csv
find ${T}_prof -name "*kernel_trace.csv" -size +2M -delete
bash tools/gpu_pmc.sh ${TAG}_pmc > ${T}_pmc.log 2>&1; tail -3 ${T}_pmc.log
rm -rf ${T}_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d ${T}_sq -- python3 bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-profile-pass --no-secondary > /dev/null 2> ${T}_sq.err
python tools/sq_summary.py ${T}_sq "k_freq_c64=k_freq<float" "k_time_mid_c64=k_time<float, 256, 16, 16, 1" > ${T}_sq.txt; cat ${T}_sq.txt
find ${T}_sq -name "*.csv" -size +1M -delete
for what in "c1:tools/c1_prof.py" "adaptive:tools/adaptive_prof.py" "sos:tools/sos_prof.py"; do
  name=${what%%:*}; script=${what#*:}
  rm -rf ${T}_${name}prof
  rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_${name}prof -- python3 $script > ${T}_${name}_run.txt 2> ${T}_${name}prof.err
  find ${T}_${name}prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} ${T}_${name}_kernel_stats.csv; head -5 ${T}_${name}_kernel_stats.csv | cut -c1-200
  find ${T}_${name}prof -name "*kernel_trace.csv" -size +1M -delete
done
bash tools/gpu_pmc_c1.sh ${TAG}_c1_pmc > ${T}_c1_pmc.log 2>&1; tail -4 ${T}_c1_pmc.log
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${TAG}_c1_$C
  timeout 400 rocprofv3 --pmc $C --output-format csv -d gpurun_out/${TAG}_c1_$C -- python3 tools/c1_prof.py > /dev/null 2> gpurun_out/${TAG}_c1_$C.err; echo "C1 $C rc=$?"
done
python tools/pmc_summary.py gpurun_out/${TAG}_c1_FETCH_SIZE gpurun_out/${TAG}_c1_WRITE_SIZE ${T}_c1_pmc_traffic.json | tail -24
find gpurun_out/${TAG}_c1_FETCH_SIZE gpurun_out/${TAG}_c1_WRITE_SIZE -name "*.csv" -size +1M -delete
python tools/cfg_times.py > ${T}_cfg_times.txt 2>&1; cat ${T}_cfg_times.txt
{ PREC=c128 POL=1 STEPS=100 python tools/step_time.py "C1, one polarisation (one lane: its kernels alone on the chip)";
  PREC=c128 POL=2 STEPS=100 python tools/step_time.py "C1 (two lanes)";
  PREC=c128 POL=2 FIELDS=2 STEPS=100 python tools/step_time.py "C1, two fields resident (four lanes)";
  python tools/attic/adapt_long.py; } > ${T}_c1_alone.txt 2>&1; cat ${T}_c1_alone.txt
python3 tools/big_n_profile.py > gpurun_out/r06_big_n.txt 2>&1; cat gpurun_out/r06_big_n.txt
python3 tools/split_check.py full > gpurun_out/r06_split_check.txt 2>&1; cat gpurun_out/r06_split_check.txt
python tools/adaptive_capture_time.py > gpurun_out/r06_adaptive_capture.txt 2>&1; tail -7 gpurun_out/r06_adaptive_capture.txt
python tools/capture_time.py > /dev/null 2>&1; cp gpurun_out/r05_capture_time.txt ${T}_capture_time.txt 2>/dev/null; cat ${T}_capture_time.txt
python tests/diag/lane_stability.py 200 > /dev/null 2>&1; cp gpurun_out/lane_stability.txt ${T}_lane_stability.txt; tail -2 ${T}_lane_stability.txt
SSFM_LANE_POOL_OFF=1 python tests/diag/lane_stability.py 60 > /dev/null 2>&1; cp gpurun_out/lane_stability.txt ${T}_lane_stability_fresh_pairs.txt; tail -2 ${T}_lane_stability_fresh_pairs.txt
{ for r in 1 2 3; do echo "== round $r"; python3 tools/filter_shapes.py; done; } > ${T}_sos_shapes.txt 2>&1; tail -6 ${T}_sos_shapes.txt
{ echo "== tests/diag/fuzz_many.py 400 2026"; timeout 1500 python tests/diag/fuzz_many.py 400 2026 2>&1 | grep -v Warning;
  echo "== tests/diag/fuzz_many.py 400 7"; timeout 1500 python tests/diag/fuzz_many.py 400 7 2>&1 | grep -v Warning;
  echo "== tests/diag/fuzz_filters.py"; timeout 600 python tests/diag/fuzz_filters.py 2>&1;
  echo "== tests/diag/fuzz_misc.py"; timeout 600 python tests/diag/fuzz_misc.py 2>&1; } > ${T}_fuzz.txt 2>&1; grep -E "cases \(seed|beyond|violation" ${T}_fuzz.txt | head
{ python3 tests/diag/capture_stress.py 120 5 | tail -3; } > ${T}_capture_stress.txt 2>&1; tail -2 ${T}_capture_stress.txt
