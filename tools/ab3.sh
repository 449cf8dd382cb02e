#!/bin/bash
# tools/ab3.sh [rounds] lib1.so lib2.so ... -- interleaved bench.py runs of several builds of the
# library in one gpurun call (same box, same clocks); one line per run with the rate and per-kernel times
R=$1; shift
for r in $(seq $R); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    KYBER_HIP_LIBRARY=$PWD/$lib timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-measurements $AB_ARGS > gpurun_out/ab3_${tag}_$r.json 2>/dev/null || exit 1
    python3 - "$tag" "$r" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab3_%s_%s.json" % (sys.argv[1], sys.argv[2])))
print("%-22s %s %6.2f M/s " % (sys.argv[1], sys.argv[2], d["value"] / 1e6) +
      " ".join("%s=%.3f" % (k.replace("k_", ""), v["avg_ms"]) for k, v in d["kernels"].items()))
PY
  done
done
