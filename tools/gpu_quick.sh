cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1l; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "sfc or separable" 2>&1 | tail -5
timeout 200 python tools/bench_sfc.py 2>&1 | tee gpurun_out/r1l/bench_sfc.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/r1l/pmc1 -o p1 -- python tools/bench_sfc.py > gpurun_out/r1l/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH --output-format csv -d gpurun_out/r1l/pmc2 -o p2 -- python tools/bench_sfc.py > gpurun_out/r1l/pmc2.log 2>&1
