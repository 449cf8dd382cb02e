#!/bin/bash
# tools/lib_ab.sh <workload> <rounds> <kernel-name-substring> <variant>... -- interleaved bench.py runs of library
# variants (tools/build_variant.sh) on one box: value and the average time of the kernels whose name contains the substring
W=$1; R=$2; K=$3; shift 3
for r in $(seq $R); do
for v in "$@"; do
  KYBER_HIP_LIBRARY=$PWD/kyber_amd/variants/libkyber_hip_$v.so timeout -k 10 200 python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-side-measurements > gpurun_out/ab_$v.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('%-12s %7.2f M/s ' % ('$v', d['value']/1e6), ' '.join('%s %.3f' % (k, x['avg_ms']) for k, x in d['kernels'].items() if '$K' in k))"
done; done
