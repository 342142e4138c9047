set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_t}
mkdir -p $OUT
for v in "" "EQF_NO_DEFER_WGRAD=1" "EQF_NO_GATE_FUSION=1"; do
  echo "=== $v" >> $OUT/ab.txt
  env $v timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -x -s -k "second_order_gradients and SMALL_L2" 2>&1 | grep -E "worst|passed|failed|AssertionError: \(" >> $OUT/ab.txt
done
cat $OUT/ab.txt
