cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2d; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
EQF_BENCH_DEVICE=0 EQF_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -4 | cut -c1-400
