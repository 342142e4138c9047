cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_ak; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sfcx.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest.txt
timeout 300 python tools/bench_sfc.py > $O/bench_sfc.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/pytest.txt; grep -h "sfcx mode [012] *bwd_weight" $O/bench_sfc.txt
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], [(k['kernel'], round(k['avg_launch_ms'],4)) for k in [d['roofline']]+d['roofline']['others'][:3]])"
