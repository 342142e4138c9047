cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2x; export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r2x/bench.json
python -c "
import json; d=json.load(open('gpurun_out/r2x/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']); print(d['north_star_kernels'].get('radial_mlp'))"
