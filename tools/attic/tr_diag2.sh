run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.2f'%d['us_per_ssfm_step'])"; }
B="bench.py --gpus 1 --steps 4 --warmup 1 --cpu-steps 0 --no-profile-pass"
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
export SSFM_GRAPH=0
echo -n "plain: "; python $B 2>/dev/null | run
echo -n "torchrun, no dist init: "; BENCH_DIST_BACKEND=none $TR --master-port 29541 $B 2>/dev/null | run
echo -n "torchrun, gloo: "; BENCH_DIST_BACKEND=gloo $TR --master-port 29542 $B 2>/dev/null | run
echo -n "torchrun, nccl: "; $TR --master-port 29543 $B 2>/dev/null | run
echo -n "plain python + RANK env + nccl: "; RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 python $B 2>&1 | tail -1 | run
env | grep -i -E "^(HSA|HIP|GPU|ROC|AMD|NCCL|RCCL|OMP)" | head -20
