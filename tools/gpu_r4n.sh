set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_n}
mkdir -p $OUT
timeout 300 python tools/host_profile.py md17_l2 10 > $OUT/host_md17_l2.txt 2>&1
grep -v "^/opt" $OUT/host_md17_l2.txt | head -14
timeout 300 python tools/host_profile.py qm9 10 > $OUT/host_qm9.txt 2>&1
grep -v "^/opt" $OUT/host_qm9.txt | head -12
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["spread"]["values"])
for c in d.get("configs",[]): print({k:c.get(k) for k in ("workload","matrix_mode","value","ms_per_step","error")})
PY
