cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r2v; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $GRAFT_REPO_ROOT/tools/bench_sfc.py 25354 > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $GRAFT_REPO_ROOT/tools/bench_sfc.py 25354 > $OUT/write.log 2>&1; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
find $OUT -name "*counter_collection.csv" | head; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
python tools/pmc_traffic.py $OUT $OUT/pmc_dominant.json
