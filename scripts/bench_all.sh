#!/bin/bash
# every bench workload once on one GPU (JSON lines under gpurun_out/bench_all/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/bench_all
for wl in grch38 paired150 chr1 config2 config2-align; do
  python3 $ROOT/bench.py --steps 5 --warmup 2 --workload $wl ${BENCH_EXTRA:-} 2> $ROOT/gpurun_out/bench_all/$wl.err | tail -1 > $ROOT/gpurun_out/bench_all/$wl.json
  python3 -c "import json,sys; d=json.load(open('$ROOT/gpurun_out/bench_all/$wl.json')); print('$wl', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['kernels_ms_per_step'].items() if v>0.1}, d.get('cpu_baseline',{}).get('parity_aggregate'))"
done
