cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2z; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "qm9_forward_backward" 2>&1 | tail -4
timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r2z/bench.json
python -c "
import json; d=json.load(open('gpurun_out/r2z/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['final_loss'])"
