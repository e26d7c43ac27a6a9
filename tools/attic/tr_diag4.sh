run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.2f'%d['us_per_ssfm_step'])"; }
B="bench.py --gpus 1 --steps 4 --warmup 1 --cpu-steps 0 --no-profile-pass"
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
echo -n "plain auto: "; python $B 2>/dev/null | run
echo -n "plain eager: "; SSFM_GRAPH=0 python $B 2>/dev/null | run
echo -n "torchrun nccl eager: "; SSFM_GRAPH=0 $TR --master-port 29561 $B 2>/dev/null | run
echo -n "torchrun nccl auto: "; $TR --master-port 29562 $B 2>/dev/null | run
