cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_l3; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_sfcx.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest.txt
timeout 100 python bench.py --workload md17_l3 --no-cpu-baseline --no-sub-records --repeats 1 > $O/bench_md17_l3.json 2> $O/err.txt
timeout 100 python tools/bench_sfc.py 2>&1 | grep "sfcx mode 0" > $O/bench_sfc.txt
cat $O/pytest.txt; cat $O/bench_sfc.txt
python -c "
import json; d=json.load(open('$O/bench_md17_l3.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], round(r['avg_launch_ms'],4), [(o['kernel'], round(o['avg_launch_ms'],4)) for o in r['others'][:3]])"
