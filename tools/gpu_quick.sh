cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1d; export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/r1d/counters.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r1d/pmc1 -o p1 -- python tools/bench_sfc.py > gpurun_out/r1d/pmc1.log 2>&1
tail -3 gpurun_out/r1d/pmc1.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d gpurun_out/r1d/pmc2 -o p2 -- python tools/bench_sfc.py > gpurun_out/r1d/pmc2.log 2>&1
tail -3 gpurun_out/r1d/pmc2.log
find gpurun_out/r1d -name "*.csv" | head; du -sh gpurun_out/r1d
