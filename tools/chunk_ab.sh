for r in 1 2 3; do
for c in 32768 65536 131072 262144 524288; do
  timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-measurements --chunk $c > gpurun_out/chunk_$c.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/chunk_$c.json')); print($c, round(d['value']/1e6,2), {k.replace('k_',''):round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done; done
