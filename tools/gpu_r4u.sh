set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_u}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemmx.py tests/test_gpu_sfcx.py tests/test_gpu_model.py -m gpu -q -x -k "not l3_full_size and not variants and not bessel" > $OUT/pytest_sel.txt 2>&1; echo "rc=$?" >> $OUT/pytest_sel.txt
grep -E "^FAILED|^ERROR|passed|failed|rc=|Error" $OUT/pytest_sel.txt | head -20
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
EQF_NO_GATE_FUSION=1 timeout 300 python bench.py --no-cpu-baseline --no-sub-records > $OUT/bench_nogate.json 2> $OUT/bench_nogate.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["spread"]["values"])
for c in d.get("configs",[]): print({k:c.get(k) for k in ("workload","matrix_mode","value","ms_per_step","error")})
d=json.load(open("$OUT/bench_nogate.json"))
print("no gate fusion:", d["value"], d["ms_per_step"], d["spread"]["values"])
PY
