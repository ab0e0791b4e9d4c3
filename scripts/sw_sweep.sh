#!/bin/bash
for v in "BWT_SEARCH_NT=99" "BWT_SEARCH_NT=8" "BWT_SEARCH_NT=10" "BWT_SEARCH_NT=12"; do
  echo "== $v"
  env $v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,1),'Mreads/s', {k:round(v,2) for k,v in d['kernels_ms_per_step'].items() if k in ('fm_search','cluster_small','sw_extend')})"
done
