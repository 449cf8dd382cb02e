#!/bin/bash
# tools/ab_bench.sh <libA.so> <libB.so> [rounds=3] -- interleaved A/B runs of
# bench.py in one gpurun call (same box, same clocks); prints each run's rate
# and per-kernel times so that two builds of the library can be compared.
set -e
A=$1; B=$2; R=${3:-3}; shift; shift; shift || true   # remaining arguments go to bench.py
for r in $(seq $R); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    KYBER_HIP_LIBRARY=$lib timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/ab_$v$r.json 2>/dev/null
    python3 - "$v" "$r" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s%s.json" % (sys.argv[1], sys.argv[2])))
print(sys.argv[1], sys.argv[2], "%.2f M/s" % (d["value"] / 1e6),
      " ".join("%s=%.3f" % (k.replace("k_", ""), v.get("avg_ms", v.get("ms"))) for k, v in d["kernels"].items()))
PY
  done
done
