set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04_b
mkdir -p $OUT
timeout 120 python tools/e3_head_bisect.py e3_missing_0o > $OUT/e3_bisect_missing.txt 2>&1
timeout 120 python tools/e3_head_bisect.py e3_complete > $OUT/e3_bisect_complete.txt 2>&1
cat $OUT/e3_bisect_complete.txt | grep -v "^/opt" | head -150
