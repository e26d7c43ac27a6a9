run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.2f'%d['us_per_ssfm_step'])"; }
echo -n "plain eager: "; python bench.py --steps 4 --warmup 1 --cpu-steps 0 --no-profile-pass 2>/dev/null | run
echo -n "RANK env (nccl init, no torchrun) eager: "; RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --steps 4 --warmup 1 --cpu-steps 0 --no-profile-pass 2>/dev/null | run
echo -n "OMP_NUM_THREADS=1 plain eager: "; OMP_NUM_THREADS=1 python bench.py --steps 4 --warmup 1 --cpu-steps 0 --no-profile-pass 2>/dev/null | run
echo -n "torchrun eager lanes=1: "; SSFM_LANES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 4 --warmup 1 --cpu-steps 0 --no-profile-pass 2>/dev/null | run
echo -n "torchrun eager lanes=2: "; python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --steps 4 --warmup 1 --cpu-steps 0 --no-profile-pass 2>/dev/null | run
