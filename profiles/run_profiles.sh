#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are checked against.  Run on the GPU box:
#   gpurun -- 'bash profiles/run_profiles.sh r1 grch38'
# Writes under gpurun_out/prof_<tag>/ ; profiles/summarize.py turns that into the committed summaries.
set -u
TAG=${1:-r1}
WL=${2:-grch38}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$WL -o trace -- \
    python3 $ROOT/bench.py --steps 30 --warmup 2 --workload $WL --no-cpu-baseline --no-pipeline > $OUT/$WL.bench.json 2> $OUT/$WL.err || exit 1
[ -n "${STATS_ONLY:-}" ] && exit 0
# HBM traffic counters, separate passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2), kernel-trace only
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${WL}_pmc_$C -o pmc -- \
      python3 $ROOT/bench.py --steps 2 --warmup 1 --workload $WL --no-cpu-baseline --no-pipeline > $OUT/${WL}_pmc_$C.bench.json 2> $OUT/${WL}_pmc_$C.err || exit 1
done
ls $OUT
