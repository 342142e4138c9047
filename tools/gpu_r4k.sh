set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_k}
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py > $OUT/pytest_all.txt 2>&1; echo "rc=$?" >> $OUT/pytest_all.txt
grep -E "^FAILED|passed|failed|rc=|Error" $OUT/pytest_all.txt | head -30
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s > $OUT/pytest_fullsize.txt 2>&1; echo "rc=$?" >> $OUT/pytest_fullsize.txt
grep -E "128 molecules|^FAILED|passed|failed|rc=" $OUT/pytest_fullsize.txt | head
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["spread"]["values"])
for c in d.get("configs",[]): print({k:c.get(k) for k in ("workload","matrix_mode","value","ms_per_step","error")})
PY
