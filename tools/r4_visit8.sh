set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for mode in "" "giveup"; do
  echo "=== order: ${mode:-plain}"
  python tools/attic/order_dependence_probe.py $mode 2>&1 | tail -4
done
echo "=== with the runtime's dispatch log"
for mode in plain giveup; do
  AMD_LOG_LEVEL=4 python tools/attic/order_dependence_probe.py $( [ $mode = giveup ] && echo giveup ) > gpurun_out/r04_probe_$mode.out 2> /tmp/probe_$mode.log
  tail -2 gpurun_out/r04_probe_$mode.out
  grep -c "HWq" /tmp/probe_$mode.log
  python tools/attic/hwq_of_lanes.py /tmp/probe_$mode.log | tee gpurun_out/r04_probe_hwq_$mode.txt
  grep -m3 "HWq" /tmp/probe_$mode.log | cut -c1-300
done
echo "=== more hardware queues"
GPU_MAX_HW_QUEUES=8 python tools/attic/order_dependence_probe.py giveup 2>&1 | tail -2
