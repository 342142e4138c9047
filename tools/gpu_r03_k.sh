cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_k; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_overlap.json 2> $O/bench_overlap.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap > $O/bench_nooverlap.json 2> $O/bench_nooverlap.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_overlap2.json 2> $O/bench_overlap2.err
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sfcx.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest.txt
for m in overlap nooverlap overlap2; do python -c "
import json; d=json.load(open('$O/bench_$m.json')); print('$m', d['value'], d['ms_per_step'], [(k['kernel'], round(k['avg_launch_ms'],4)) for k in [d['roofline']]+d['roofline']['others'][:3]])"; done; cat $O/pytest.txt
