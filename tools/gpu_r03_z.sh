cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_z; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention or alpha" 2>&1 | tail -4 > $O/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/pytest.txt
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['north_star_kernels']['scatter'])"
