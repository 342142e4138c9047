cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for w in md17_l3 oc20; do
  O=gpurun_out/r03_wlprof_$w; mkdir -p $O
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-sub-records --repeats 1 > $O/bench.json 2> $O/err.txt
  DB=$(find $O/prof -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB --csv $O/kernel_stats.csv --top 40 > $O/kernel_stats.txt
  rm -rf $O/prof
  head -14 $O/kernel_stats.txt | cut -c1-70,100-150
done
