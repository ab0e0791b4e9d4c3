#!/bin/bash
# SQ counters (VALU / LDS / wait cycles) of the seed, cluster and extend kernels; usage: pmc_sq.sh [workload]
WL=${1:-grch38}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/a -o pmc -- \
   python3 $ROOT/bench.py --steps 1 --warmup 1 --workload $WL --no-cpu-baseline --no-pipeline > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -o pmc -- \
   python3 $ROOT/bench.py --steps 1 --warmup 1 --workload $WL --no-cpu-baseline --no-pipeline > $OUT/b.json 2> $OUT/b.err
python3 - <<'PY'
import csv, collections, glob, os
root=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_sq"
for d in ("a","b"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob(root+"/"+d+"/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if ("k_sw_pk" in k) or k.startswith("k_search") or k.startswith("k_cluster_small") or ("k_seed_tasks" in k) or ("k_seed_fast" in k) or ("k_sw_prefilter" in k) or ("k_sw_band" in k):
                agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k in agg:
        print(k, {c: "%.4g"%(v/n[(k,c)]) for c,v in agg[k].items()})
PY
