# round 4, call A: new tests (one-rank RCCL, planner fallback), E(3) aux-head bisect, new bench line with sub-records
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04_a
mkdir -p $OUT
nproc > $OUT/nproc.txt
timeout 300 python -m pytest tests/test_gpu_parallel.py -m gpu -x -q > $OUT/pytest_parallel.txt 2>&1; echo "rc=$?" >> $OUT/pytest_parallel.txt
tail -5 $OUT/pytest_parallel.txt
timeout 300 python -m pytest tests/test_gpu_sfcx.py -m gpu -x -q -k "wide_l2" > $OUT/pytest_wide.txt 2>&1; echo "rc=$?" >> $OUT/pytest_wide.txt
tail -5 $OUT/pytest_wide.txt
timeout 120 python tools/e3_head_bisect.py e3_missing_0o > $OUT/e3_bisect_missing.txt 2>&1
timeout 120 python tools/e3_head_bisect.py e3_complete > $OUT/e3_bisect_complete.txt 2>&1
grep -n "output\|<--" $OUT/e3_bisect_missing.txt | head -20
grep -n "output\|<--" $OUT/e3_bisect_complete.txt | head -20
timeout 400 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_a/bench.json"))
print(d["value"], d["ms_per_step"], d["spread"]["values"])
print({k:d["roofline"][k] for k in ("kernel","avg_launch_ms","achieved","peak","frac","frac_of_fp32_peak","mfma_busy","traffic_over_algorithmic")})
for c in d.get("configs",[]): print({k:c.get(k) for k in ("workload","matrix_mode","value","ms_per_step","error","wall_s")})
PY
