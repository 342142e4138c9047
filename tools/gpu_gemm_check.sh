set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-rX}
mkdir -p $OUT
timeout 120 python tools/gemm_shapes.py split 2>&1 | grep "node\|edge" > $OUT/shapes.txt
cat $OUT/shapes.txt
timeout 400 python -m pytest tests/test_gpu_gemmx.py tests/test_gpu_ops.py -m gpu -q -x > $OUT/pytest_sel.txt 2>&1; echo "rc=$?" >> $OUT/pytest_sel.txt
grep -E "^FAILED|^ERROR|passed|failed|rc=|Error" $OUT/pytest_sel.txt | head
timeout 300 python bench.py --no-sub-records --repeats 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for o in [d["roofline"]]+d["roofline"].get("others",[]): print(o["kernel"], o["launches"], round(o["avg_launch_ms"]*1000,1))
PY
