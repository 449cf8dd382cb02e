#!/bin/bash
# tools/pmc_pass.sh <tag> <counters...> -- one rocprofv3 --pmc pass over a short bench run
# (separate run per counter group, as the MI355X guide prescribes; no trace domains combined)
set -e
tag=$1; shift
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 1 --warmup 1 --batch 524288 --no-cpu-baseline > gpurun_out/pmc_$tag.json 2> gpurun_out/pmc_$tag.err
