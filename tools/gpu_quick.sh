cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1w; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r1w/pmc1 -o p1 -- python tools/bench_sfc.py > gpurun_out/r1w/pmc1.log 2>&1
tail -2 gpurun_out/r1w/pmc1.log
