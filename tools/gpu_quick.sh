cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2y; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_optim.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r2y/bench.json
python -c "
import json; d=json.load(open('gpurun_out/r2y/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['final_loss'])"
