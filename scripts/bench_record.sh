#!/bin/bash
# the bench lines committed under profiles/<tag>/ (default run incl. the CPU baseline leg)
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bench_$TAG
mkdir -p $OUT
python3 $ROOT/bench.py 2> $OUT/grch38.err | tail -1 > $OUT/bench_grch38_n1.json
python3 $ROOT/bench.py --workload paired150 2> $OUT/paired150.err | tail -1 > $OUT/bench_paired150_n1.json
python3 $ROOT/bench.py --workload chr1 2> $OUT/chr1.err | tail -1 > $OUT/bench_chr1_n1.json
python3 $ROOT/bench.py --workload config2 2> $OUT/config2.err | tail -1 > $OUT/bench_config2_search.json
python3 $ROOT/bench.py --workload config2-align 2> $OUT/config2a.err | tail -1 > $OUT/bench_config2_align.json
for f in $OUT/*.json; do python3 -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']/1e6,1), d['roofline']['kernel'], round(d['roofline']['frac'],3), d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('parity_aggregate',{}).get('equal'), d.get('cpu_baseline',{}).get('parity_sample'))"; done
