#!/bin/bash
# experiment driver: SW pass-1 variants on the chr1-scale workload
for v in "BWT_SW_LDS=1" "BWT_SW_LDS=0 BWT_SW_WAVES_PER_CU=8" "BWT_SW_LDS=0 BWT_SW_WAVES_PER_CU=16" "BWT_SW_LDS=0 BWT_SW_WAVES_PER_CU=24"; do
  echo "== $v"
  env $v python bench.py --workload chr1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,1),'Mreads/s', {k:round(v,2) for k,v in d['kernels_ms_per_step'].items() if k.startswith('sw')})"
done
