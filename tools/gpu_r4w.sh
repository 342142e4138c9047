set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_w}
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_gemmx.py -m gpu -q -x > $OUT/pytest_gemmx.txt 2>&1; echo "rc=$?" >> $OUT/pytest_gemmx.txt
grep -E "^FAILED|passed|failed|rc=|Error" $OUT/pytest_gemmx.txt | head
timeout 300 python tools/gemm_shapes.py split,bf16 > $OUT/gemm_shapes.txt 2>&1
grep "edge" $OUT/gemm_shapes.txt
timeout 300 python bench.py --no-cpu-baseline --no-sub-records > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json
d=json.load(open('$OUT/bench.json'))
print(d['value'], d['ms_per_step'], d['spread']['values']); print(d['north_star_kernels']['radial_mlp'])"
