run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.2f'%d['us_per_ssfm_step'])"; }
B="bench.py --gpus 1 --steps 4 --warmup 1 --cpu-steps 0 --no-profile-pass"
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
export SSFM_GRAPH=0
for Q in 2 4 8 16; do echo -n "torchrun nccl eager GPU_MAX_HW_QUEUES=$Q: "; GPU_MAX_HW_QUEUES=$Q $TR --master-port 2955$Q $B 2>/dev/null | run; done
echo -n "plain GPU_MAX_HW_QUEUES=8: "; GPU_MAX_HW_QUEUES=8 python $B 2>/dev/null | run
