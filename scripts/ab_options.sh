# one-off A/B runs of library options on one workload (resident steps): usage r3_exp.sh <workload> "<ENV=..>" "<ENV=..>" ...
WL=$1; shift
for cfg in "$@"; do
  env $cfg python bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-pipeline 2> gpurun_out/exp_$WL.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$WL', '$cfg', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['kernels_ms_per_step'].items() if v>0.08}, 'handed_back', d['counts_per_step']['n_handed_back'])"
  grep "handed back\|cluster kernels" gpurun_out/exp_$WL.err | tail -2
done
