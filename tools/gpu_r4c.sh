set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04_c
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_e3.py tests/test_gpu_oc20_heads.py -m gpu -q -s > $OUT/pytest_e3_heads.txt 2>&1; echo "rc=$?" >> $OUT/pytest_e3_heads.txt
grep -n "rel err\|worst\|passed\|failed\|Error\|rc=" $OUT/pytest_e3_heads.txt | head -60
