set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_q}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemmx.py tests/test_gpu_parallel.py tests/test_gpu_model.py tests/test_optim.py -m gpu -q -x -k "not l3_full_size and not second_order and not variants" > $OUT/pytest_sel.txt 2>&1; echo "rc=$?" >> $OUT/pytest_sel.txt
grep -E "^FAILED|^ERROR|passed|failed|rc=|Error" $OUT/pytest_sel.txt | head -20
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["spread"]["values"])
for o in d["roofline"]["others"]: print({k:o[k] for k in ("kernel","launches","avg_launch_ms")})
for c in d.get("configs",[]): print({k:c.get(k) for k in ("workload","matrix_mode","value","ms_per_step","error")})
PY
