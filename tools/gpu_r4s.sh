set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_s}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_sfcx.py -m gpu -q -x -s -k "gate_folded" > $OUT/pytest_gate.txt 2>&1; echo "rc=$?" >> $OUT/pytest_gate.txt
grep -E "gate folded|^FAILED|passed|failed|rc=|Error" $OUT/pytest_gate.txt | head
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_sfcx.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not l3_full_size and not variants and not bessel" > $OUT/pytest_sel.txt 2>&1; echo "rc=$?" >> $OUT/pytest_sel.txt
grep -E "^FAILED|^ERROR|passed|failed|rc=|Error" $OUT/pytest_sel.txt | head -20
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["spread"]["values"])
for o in [d["roofline"]]+d["roofline"]["others"][:3]: print({k:o[k] for k in ("kernel","launches","avg_launch_ms")})
for c in d.get("configs",[]): print({k:c.get(k) for k in ("workload","matrix_mode","value","ms_per_step","error")})
PY
