set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_r}
mkdir -p $OUT
bash tools/gpu_profile.sh ${1:-r04_r} > $OUT/profile_script.log 2>&1
cat $OUT/pmc_summary.txt
head -30 $OUT/kernel_stats.txt | cut -c1-100,112-160
tail -1 $OUT/kernel_stats.txt
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["spread"]["values"])
print({k:d["roofline"][k] for k in ("kernel","avg_launch_ms","achieved","peak","frac","mfma_busy","traffic_over_algorithmic")})
for c in d.get("configs",[]): print({k:c.get(k) for k in ("workload","matrix_mode","value","ms_per_step","error")})
print(d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"))
PY
