cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2p; export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
EQF_BENCH_DEVICE=0 EQF_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
