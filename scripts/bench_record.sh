#!/bin/bash
# the bench lines committed under profiles/<tag>/ (default runs incl. the CPU baseline leg and its parity checks)
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bench_$TAG
mkdir -p $OUT
run() { name=$1; shift; python3 $ROOT/bench.py "$@" 2> $OUT/$name.err | tail -1 > $OUT/bench_$name.json; echo "[bench_record] $name done"; }
run grch38_n1 --steps 20 --warmup 5
run grch38_repeats_n1 --workload grch38-repeats --steps 10 --warmup 3
run paired150_n1 --workload paired150 --steps 10 --warmup 3
run chr1_n1 --workload chr1 --steps 20 --warmup 5
run config2_search --workload config2 --steps 20 --warmup 5
run config2_align --workload config2-align --steps 20 --warmup 5
for f in $OUT/*.json; do python3 -c "
import json
d=json.load(open('$f'))
print('$f'.split('/')[-1], round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms/step; resident', round(d.get('resident',{}).get('reads_per_s',0)/1e6,1), d['roofline']['kernel'], round(d['roofline']['frac'],3), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('parity_aggregate',{}).get('equal'), d.get('cpu_baseline',{}).get('parity_sample'))"; done
