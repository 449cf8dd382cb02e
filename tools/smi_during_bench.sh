#!/bin/bash
# tools/smi_during_bench.sh -- samples the board's power and clocks while bench.py runs (reading only)
python3 bench.py --steps 600 --warmup 5 --no-cpu-baseline --no-side-measurements > gpurun_out/smi_bench.json 2>/dev/null &
BP=$!
for i in $(seq 40); do
  printf "t=%s " "$(date +%s.%N | cut -c1-14)"
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk" | sed 's/^GPU\[[0-9]*\][ \t]*: //' | tr '\n' ';'
  echo
  sleep 0.3
  kill -0 $BP 2>/dev/null || break
done
wait $BP
python3 -c "import json; d=json.load(open('gpurun_out/smi_bench.json')); print(d['value']/1e6, 'M/s')"
