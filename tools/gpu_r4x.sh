set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_x}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_optim.py tests/test_gpu_parallel.py tests/test_gpu_gemmx.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not bf16_mode" > $OUT/pytest_sel.txt 2>&1; echo "rc=$?" >> $OUT/pytest_sel.txt
grep -E "^FAILED|^ERROR|passed|failed|rc=|Error" $OUT/pytest_sel.txt | head -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["spread"]["values"], d["config"]["final_loss"])
for o in d["roofline"]["others"]: print({k:o[k] for k in ("kernel","launches","avg_launch_ms")})
for c in d.get("configs",[]): print({k:c.get(k) for k in ("workload","matrix_mode","value","ms_per_step","error")})
PY
