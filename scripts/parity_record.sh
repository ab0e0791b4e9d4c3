#!/bin/bash
# the parity-at-scale records of a round (complete answers, read by read, against the CPU oracle): usage parity_record.sh <tag>
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/parity_$TAG
mkdir -p $OUT
run() { name=$1; shift; python3 $ROOT/scripts/parity_at_scale.py "$@" > $OUT/$name.json 2> $OUT/$name.err || tail -3 $OUT/$name.err; python3 -c "
import json
d=json.load(open('$OUT/$name.json')); print('$name', d['reads'], 'reads', d['hits'], 'hits, mismatching', d['mismatching_reads'])"; }
run parity_at_scale --reads 4000000
run parity_at_scale_repeats --reads 3000000 --repeats 300 --max-subs 5
run parity_at_scale_150bp --reads 2000000 --length 150
run parity_at_scale_spectrum --reads 1500000 --spectrum
run parity_at_scale_spectrum_150bp --reads 1000000 --spectrum --length 150
