# N>1 code path of bench.py / eval_sharded.py on a 1-GPU box: 2 ranks share cuda:0, collectives over gloo
export PMCE_DIST_BACKEND=gloo PMCE_BENCH_SHARE_GPU=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 2>&1 | grep -v amdgpu | tail -3
