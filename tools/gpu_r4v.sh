set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_v}
mkdir -p $OUT
for v in fused nogate; do
  if [ $v = nogate ]; then export EQF_NO_GATE_FUSION=1; else unset EQF_NO_GATE_FUSION; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o $v -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub-records --repeats 1 > $OUT/bench_$v.json 2> $OUT/prof_$v.err
  DB=$(find $OUT/prof_$v -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB --csv $OUT/kernel_stats_$v.csv --top 80 > $OUT/kernel_stats_$v.txt
  rm -rf $OUT/prof_$v
  tail -1 $OUT/kernel_stats_$v.txt
  grep -E "sfcx_|gate_" $OUT/kernel_stats_$v.txt | cut -c1-60,100-160
done
