set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_g}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_gemmx.py -m gpu -x -q -s > $OUT/pytest_gemmx.txt 2>&1; echo "rc=$?" >> $OUT/pytest_gemmx.txt
grep -v "^/opt" $OUT/pytest_gemmx.txt | tail -45
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q > $OUT/pytest_model_ops.txt 2>&1; echo "rc=$?" >> $OUT/pytest_model_ops.txt
tail -5 $OUT/pytest_model_ops.txt
timeout 300 python bench.py --no-cpu-baseline --no-sub-records > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_g/bench.json"))
print(d["value"], d["ms_per_step"], d["spread"]["values"])
for o in [d["roofline"]]+d["roofline"]["others"]: print({k:o[k] for k in ("kernel","launches","avg_launch_ms","achieved","frac")})
PY
