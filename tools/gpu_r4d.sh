set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04_d
mkdir -p $OUT
timeout 300 python tools/bench_sfcx.py 25354 0,1 --dm > $OUT/bench_sfcx.txt 2>&1
cat $OUT/bench_sfcx.txt
timeout 600 python -m pytest tests/test_gpu_sfcx.py -m gpu -x -q > $OUT/pytest_sfcx.txt 2>&1; echo "rc=$?" >> $OUT/pytest_sfcx.txt
tail -5 $OUT/pytest_sfcx.txt
