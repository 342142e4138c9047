cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_ad; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention or alpha" 2>&1 | tail -3 > $O/pytest.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err
DB=$(find $O/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --csv $O/kernel_stats.csv --top 80 > $O/kernel_stats.txt
rm -rf $O/prof
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/pytest.txt; grep "attn" $O/kernel_stats.txt | cut -c1-50,100-150
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['north_star_kernels']['scatter'])"
