set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_i}
mkdir -p $OUT
timeout 300 python tools/gemm_shapes.py split,bf16 > $OUT/gemm_shapes.txt 2>&1
grep -v "^/opt" $OUT/gemm_shapes.txt
timeout 600 python -m pytest tests/test_gpu_gemmx.py tests/test_gpu_ops.py -m gpu -q > $OUT/pytest_ops.txt 2>&1; echo "rc=$?" >> $OUT/pytest_ops.txt
grep -E "^FAILED|passed|failed|rc=" $OUT/pytest_ops.txt | head -30
timeout 300 python bench.py --no-cpu-baseline --no-sub-records > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["spread"]["values"])
for o in [d["roofline"]]+d["roofline"]["others"]: print({k:o[k] for k in ("kernel","launches","avg_launch_ms","achieved","frac")})
PY
