# usage (on the GPU box, via gpurun): bash tools/gpu_ab.sh <tag> <variant> [bench args]
# the product library and a variant build (equiformer_amd/build.py --variant) interleaved on ONE box: bench line twice each, then
# one rocprofv3 kernel-trace summary each -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; VAR=$2; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for i in 1 2; do
  for v in "" $VAR; do
    EQF_LIB_VARIANT=$v python bench.py --no-cpu-baseline --no-sub-records --repeats 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [%s] %.1f %s  %.3f ms/step' % ('$v', d['value'], d['unit'], d['ms_per_step']))"
  done
done | tee $OUT/ab.txt
for v in "" $VAR; do
  EQF_LIB_VARIANT=$v timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub-records --repeats 1 "$@" > /dev/null 2> $OUT/prof_$v.err
  DB=$(find $OUT/prof_$v -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB --top 60 > $OUT/kernel_stats_${v:-product}.txt
  rm -rf $OUT/prof_$v $OUT/prof_$v.err
done
