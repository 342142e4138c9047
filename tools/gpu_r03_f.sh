# round 3: full GPU suite with the split-precision kernels as the default + bench lines in both arithmetic modes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_f; mkdir -p $O; export TMPDIR=/tmp
timeout 200 python tools/bench_sfc.py 2>&1 | grep "sfcx mode 0\|order" > $O/bench_sfc.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_split.json 2> $O/bench_split.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --matrix-mode bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --matrix-mode fp32 > $O/bench_fp32.json 2> $O/bench_fp32.err
cat $O/bench_sfc.txt; tail -6 $O/pytest_gpu.txt; for m in split bf16 fp32; do python -c "
import json; d=json.load(open('$O/bench_$m.json')); print('$m', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done
