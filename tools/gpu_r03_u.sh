cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_u; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest.txt
python tools/gemm_node_exp.py 2>&1 | grep "exp=0" > $O/gemm_node.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/pytest.txt $O/gemm_node.txt
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], [(k['kernel'], round(k['avg_launch_ms'],4)) for k in [d['roofline']]+d['roofline']['others'][:6]]); print(d['north_star_kernels']['radial_mlp']['avg_launch_ms'])"
