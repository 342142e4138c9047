set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_o}
mkdir -p $OUT
timeout 300 python tools/host_sample.py md17_l2 30 > $OUT/sample_md17_l2.txt 2>&1
grep -v "^/opt" $OUT/sample_md17_l2.txt | head -75
