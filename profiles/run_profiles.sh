#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are checked against.  Run on the GPU box:
#   gpurun -- 'bash profiles/run_profiles.sh r1'
# Writes under gpurun_out/prof_<tag>/ ; copy the *_stats.csv / counter summaries into profiles/.
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for WL in config2 config2-align; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$WL -o trace -- \
      python3 $ROOT/bench.py --steps 10 --warmup 2 --workload $WL --no-cpu-baseline > $OUT/$WL.bench.json 2> $OUT/$WL.err || exit 1
done
# HBM traffic counters, separate passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- \
      python3 $ROOT/bench.py --steps 3 --warmup 1 --workload config2-align --no-cpu-baseline > $OUT/pmc_$C.bench.json 2> $OUT/pmc_$C.err || exit 1
done
find $OUT -name '*.csv' | head -50
