#!/bin/bash
# the headline workload under the deployment variants INTEGRATION.md talks about: smaller suffix-interval tables (HBM budget per
# index) and the N > 1 exchange step rehearsed with a world of one.  usage: deploy_variants.sh <tag>
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/deploy_$TAG
mkdir -p $OUT
one() { name=$1; shift; env "$@" > $OUT/$name.json 2> $OUT/$name.err || tail -5 $OUT/$name.err; python3 -c "
import json
d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value']/1e6,1), 'M reads/s at the host boundary,', round(d['resident']['reads_per_s']/1e6,1), 'resident; table bytes', d['config']['index_bytes']['ktable'], {k: round(v,2) for k,v in d['kernels_ms_per_step'].items() if v > 0.3}, d['counts_per_step'].get('gathered_hits'))"; }
one ktable_15_wide python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline
one ktable_15_narrow python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --ktable 15,0
one ktable_13_narrow python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --ktable 13,0
one ktable_none python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --ktable 0,0
one force_gather BWT_BENCH_FORCE_GATHER=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline
