# usage (GPU box): bash tools/pmc_quick.sh <out dir> <variant or ""> : FETCH_SIZE / WRITE_SIZE of the sfcx kernels over tools/bench_sfcx.py
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$1; V=$2; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  EQF_LIB_VARIANT=$V timeout 300 rocprofv3 --pmc $c -d $OUT/pmc_$c --output-format csv -- python tools/bench_sfcx.py 25354 0 > /dev/null 2> $OUT/pmc_$c.err
done
python - <<PY
import csv, glob, collections
v = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for n in ("sfcx_bwd_kernel", "sfcx_wgrad_kernel", "sfcy_fwd_kernel", "sfcx_fwd_kernel"):
            if n in k:
                v[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in v.items():
    f, w = c.get("FETCH_SIZE", []), c.get("WRITE_SIZE", [])
    h = len(f) // 2
    if f and w:
        print("[%s] %-20s sep_act fetch %.1f MB write %.1f MB | sep_value fetch %.1f MB write %.1f MB" % ("$V", n, 2 * 1024 * sum(f[:h]) / max(h, 1) / 1e6, 1024 * sum(w[:h]) / max(h, 1) / 1e6, 2 * 1024 * sum(f[h:]) / max(len(f) - h, 1) / 1e6, 1024 * sum(w[h:]) / max(len(w) - h, 1) / 1e6))
PY
find $OUT -name '*.csv' -size +500k -delete
