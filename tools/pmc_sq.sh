# usage (GPU box): bash tools/pmc_sq.sh <out dir> <variant or ""> [kernel substring] : SQ counters (busy / wait / instruction mix) of one
# sfc kernel over tools/bench_sfcx.py, one --pmc pass per counter group (no trace options beside --pmc)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$1; V=$2; K=${3:-sfcx_wgrad_kernel}; mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  EQF_LIB_VARIANT=$V timeout 300 rocprofv3 --pmc $grp -d $OUT/g$i --output-format csv -- python tools/bench_sfcx.py 25354 0 > /dev/null 2> $OUT/g$i.err
  tail -2 $OUT/g$i.err
done
python - <<PY
import csv, glob, collections
v = collections.defaultdict(list)
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$K" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(v.items()):
    h = len(c) // 2
    print("[%s] %-28s sep_act %14.0f | sep_value %14.0f   (%d launches)" % ("$V", n, sum(c[:h]) / max(h, 1), sum(c[h:]) / max(len(c) - h, 1), len(c)))
PY
find $OUT -name '*.csv' -size +500k -delete
