set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_j}
mkdir -p $OUT
timeout 300 python tools/gemm_shapes.py split > $OUT/gemm_shapes_direct.txt 2>&1
timeout 300 python tools/gemm_shapes.py split --no-direct > $OUT/gemm_shapes_tiled.txt 2>&1
grep -v "^/opt" $OUT/gemm_shapes_direct.txt | head -3; grep -v "^/opt" $OUT/gemm_shapes_tiled.txt | head -3
timeout 600 python -m pytest tests/test_gpu_gemmx.py tests/test_gpu_ops.py -m gpu -q > $OUT/pytest_ops.txt 2>&1; echo "rc=$?" >> $OUT/pytest_ops.txt
grep -E "^FAILED|passed|failed|rc=" $OUT/pytest_ops.txt | head -30
