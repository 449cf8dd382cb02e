#!/bin/bash
# tools/pmc_pass.sh <tag> <counters...> -- one rocprofv3 --pmc pass over a short bench run
# (separate run per counter group, as the MI355X guide prescribes; no trace domains combined).
# PMC_BENCH_ARGS overrides the bench.py arguments (default: one 2^19-instance chunk per launch).
set -e
tag=$1; shift
export TMPDIR=/tmp
ARGS=${PMC_BENCH_ARGS:---steps 1 --warmup 1 --batch 524288 --no-cpu-baseline --no-side-measurements}
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py $ARGS > gpurun_out/pmc_$tag.json 2> gpurun_out/pmc_$tag.err
