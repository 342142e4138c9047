# round-3 first device run of the split-precision SeparableFCTP kernels: reproducer, parity, micro-benchmark, bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_a; mkdir -p $O; export TMPDIR=/tmp
(hipcc --offload-arch=gfx950 -O2 tools/pk_fp32_beside_bf16_mfma.hip -o /tmp/pk 2>/dev/null && timeout 120 /tmp/pk) > $O/pk_reproducer.txt 2>&1; echo "pk rc $?" >> $O/pk_reproducer.txt
timeout 900 python -m pytest tests/test_gpu_sfcx.py -m gpu -q -s -x 2>&1 | tail -60 > $O/pytest_sfcx.txt
timeout 300 python tools/bench_sfc.py > $O/bench_sfc.txt 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_split.json 2> $O/bench_split.err
tail -3 $O/pk_reproducer.txt; tail -15 $O/pytest_sfcx.txt; grep sfcx $O/bench_sfc.txt | head -30; cat $O/bench_split.json | head -c 1500
