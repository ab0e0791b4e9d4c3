#!/bin/bash
for v in "BWT_SW_PK=1" "BWT_SW_PK=0"; do
  echo "== $v"
  env $v python bench.py --workload chr1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,1),'Mreads/s', {k:round(v,2) for k,v in d['kernels_ms_per_step'].items() if k.startswith('sw')}, d['counts_per_step']['n_hits'], d['counts_per_step']['n_traced'])"
done
