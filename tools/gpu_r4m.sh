set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_m}
mkdir -p $OUT
timeout 300 python tools/host_profile.py md17_l2 10 > $OUT/host_md17_l2.txt 2>&1
grep -v "^/opt" $OUT/host_md17_l2.txt | head -60
timeout 300 python tools/host_profile.py qm9 10 > $OUT/host_qm9.txt 2>&1
grep -v "^/opt" $OUT/host_qm9.txt | head -12
