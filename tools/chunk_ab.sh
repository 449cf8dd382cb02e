#!/bin/bash
# tools/chunk_ab.sh [workload] -- interleaved bench.py runs at several internal pass sizes (kyber_hip_set_chunk)
W=${1:-kyber768}
for r in 1 2 3 4 5; do
for c in 262144 524288; do
  timeout -k 10 200 python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-side-measurements --chunk $c > gpurun_out/chunk_$c.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/chunk_$c.json')); print('$W', $c, round(d['value']/1e6,2))"
done; done
