#!/bin/bash
# tools/ab_ntt.sh [rounds] lib1.so lib2.so ... -- interleaved `bench.py --workload ntt` runs of several builds
R=$1; shift
for r in $(seq $R); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    KYBER_HIP_LIBRARY=$PWD/$lib timeout -k 10 100 python3 bench.py --workload ntt --steps 20 --warmup 3 > gpurun_out/abntt_${tag}_$r.json 2>/dev/null || exit 1
    python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
print('%-20s'%sys.argv[2], ' '.join('%s=%.3fG'%(k.replace('poly_',''),v['polys_per_s']/1e9) for k,v in d['kernels'].items()))" gpurun_out/abntt_${tag}_$r.json $tag
  done
done
