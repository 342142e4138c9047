# final run of a round: profile (bench line + rocprofv3 kernel stats + PMC passes) first, then the whole GPU suite
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-rX}
mkdir -p $OUT
bash tools/gpu_profile.sh ${1:-rX} > $OUT/profile_script.log 2>&1
tail -5 $OUT/pmc_summary.txt
head -42 $OUT/kernel_stats.txt | cut -c1-100,112-160
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["spread"]["values"])
print({k:d["roofline"][k] for k in ("kernel","avg_launch_ms","achieved","peak","frac","mfma_busy","traffic_over_algorithmic")})
print(d["north_star_kernels"])
print(d.get("cpu_baseline",{}).get("value"))
print([(c["workload"], c["matrix_mode"], round(c["value"],1), round(c["ms_per_step"],2)) for c in d.get("configs",[])])
PY
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_all.txt 2>&1; echo "rc=$?" >> $OUT/pytest_all.txt
grep -E "^FAILED|^ERROR|passed|failed|rc=" $OUT/pytest_all.txt | head -30
