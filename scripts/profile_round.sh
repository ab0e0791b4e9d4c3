#!/bin/bash
# One workload's profile record for profiles/<tag>/ (run on the GPU box; copy gpurun_out/prof_<tag>/<wl>_* into profiles/<tag>/):
#   <wl>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the bench command (per-kernel calls / average duration)
#   <wl>_bench_under_rocprof.json   the JSON line of that same run (its hipEvent kernel times must agree with the stats)
#   <wl>_pmc_hbm.csv           FETCH_SIZE and WRITE_SIZE per kernel and launch (separate --pmc passes, MI355X_MICROARCH.md HBM section)
#   <wl>_pmc_sq.txt            SQ counters of the main kernels (two passes)
# usage: profile_round.sh <tag> <workload> [steps]
TAG=${1:-r3}; WL=${2:-grch38}; STEPS=${3:-6}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (resident steps only: under rocprofv3's tracing the pipe's streams are serialised against each other and a kernel's hipEvent
# pair then also brackets its wait for another slot's copy / kernel -- the bytes per launch and the undisturbed durations are
# what the profile is for; the default bench run times the same kernels live, inside the pipeline)
BENCH="python3 $ROOT/bench.py --steps $STEPS --warmup 2 --workload $WL --no-cpu-baseline --no-pipeline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${WL}_stats -o st -- $BENCH > $OUT/${WL}_bench_under_rocprof.json 2> $OUT/${WL}_stats.err || { tail -20 $OUT/${WL}_stats.err; exit 1; }
echo "[profile] $WL kernel stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${WL}_pmc_$C -o pmc -- $BENCH > $OUT/${WL}_pmc_$C.json 2> $OUT/${WL}_pmc_$C.err || { tail -20 $OUT/${WL}_pmc_$C.err; exit 1; }
  echo "[profile] $WL $C done"
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/${WL}_sqa -o pmc -- $BENCH > $OUT/${WL}_sqa.json 2> $OUT/${WL}_sqa.err || { tail -20 $OUT/${WL}_sqa.err; exit 1; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${WL}_sqb -o pmc -- $BENCH > $OUT/${WL}_sqb.json 2> $OUT/${WL}_sqb.err || { tail -20 $OUT/${WL}_sqb.err; exit 1; }
echo "[profile] $WL SQ passes done"
python3 $ROOT/scripts/profile_summarise.py $OUT $WL
