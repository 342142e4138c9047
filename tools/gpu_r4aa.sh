set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_ab}
mkdir -p $OUT
for ms in 4 8 16 32; do echo "== tn min steps $ms" >> $OUT/tn.txt; timeout 120 python tools/gemm_shapes.py split --tn-minsteps=$ms 2>&1 | grep "node\|edge 7" >> $OUT/tn.txt; done
cat $OUT/tn.txt
timeout 400 python -m pytest tests/test_gpu_gemmx.py tests/test_gpu_ops.py -m gpu -q -x > $OUT/pytest_sel.txt 2>&1; echo "rc=$?" >> $OUT/pytest_sel.txt
grep -E "^FAILED|^ERROR|passed|failed|rc=|Error" $OUT/pytest_sel.txt | head
timeout 300 python bench.py --no-sub-records --repeats 1 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json | cut -c1-600
