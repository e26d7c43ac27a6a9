set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r04_order_dependence.txt
for k in 0 1 2 3 4 5 6 8; do
  AMD_LOG_LEVEL=4 python tools/attic/order_dependence_probe.py dummies=$k > /tmp/out_$k.txt 2> /tmp/log_$k.txt
  { echo "=== $k other plans alive (one used high-priority stream each)"; tail -2 /tmp/out_$k.txt; python tools/attic/hwq_of_lanes.py /tmp/log_$k.txt; grep -c "Created SWq" /tmp/log_$k.txt; } >> gpurun_out/r04_order_dependence.txt
done
echo "=== 6 other plans, GPU_MAX_HW_QUEUES=8" >> gpurun_out/r04_order_dependence.txt
GPU_MAX_HW_QUEUES=8 python tools/attic/order_dependence_probe.py dummies=6 2>/dev/null | tail -2 >> gpurun_out/r04_order_dependence.txt
cat gpurun_out/r04_order_dependence.txt
