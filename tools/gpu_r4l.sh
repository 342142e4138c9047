set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_l}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_sfcx.py -m gpu -x -q > $OUT/pytest_sfcx.txt 2>&1; echo "rc=$?" >> $OUT/pytest_sfcx.txt
tail -3 $OUT/pytest_sfcx.txt
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "md17" > $OUT/pytest_md17.txt 2>&1; echo "rc=$?" >> $OUT/pytest_md17.txt
grep -E "L3 full|worst|passed|failed|rc=" $OUT/pytest_md17.txt | tail -12
timeout 200 python bench.py --workload md17_l3 --no-cpu-baseline --no-sub-records --repeats 1 > $OUT/bench_md17_l3.json 2> $OUT/bench_md17_l3.err
python -c "import json;d=json.load(open('$OUT/bench_md17_l3.json'));print('md17_l3',d['value'],d['ms_per_step'])"
timeout 200 python tools/bench_sfcx.py 25354 0 > $OUT/bench_sfcx.txt 2>&1; grep -v "^/opt" $OUT/bench_sfcx.txt
