set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_h}
mkdir -p $OUT
timeout 300 python tools/gemm_shapes.py fp32,split,bf16 > $OUT/gemm_shapes.txt 2>&1
grep -v "^/opt" $OUT/gemm_shapes.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q > $OUT/pytest_ops.txt 2>&1; echo "rc=$?" >> $OUT/pytest_ops.txt
grep -E "^FAILED|passed|failed|rc=" $OUT/pytest_ops.txt | head -30
