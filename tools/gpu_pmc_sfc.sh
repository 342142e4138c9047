# usage (on the GPU box, via gpurun): bash tools/gpu_pmc_sfc.sh <tag>   -- SQ / TA counters of the sfc and sfcx kernels
# (each --pmc set in its own pass, no trace options beside --pmc)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-rX_pmc}; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d $O/pass$i --output-format csv -- python tools/bench_sfc.py > /dev/null 2> $O/pass$i.err
done
python tools/pmc_sfc.py $O > $O/pmc_summary.txt 2>&1
find $O -name '*.csv' -size +1000k -delete
cat $O/pmc_summary.txt | cut -c1-260
