cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2k; export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r2k/bench.json; python -c "
import json; d=json.load(open('gpurun_out/r2k/bench.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['north_star_kernels'], indent=1))"
