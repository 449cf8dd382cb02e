#!/bin/bash
# tools/round_profiles.sh <rNN> -- everything the round's profiles/ entries come from, in one GPU call:
# the default bench line, the side workloads, rocprofv3 kernel-trace stats of the same commands and the
# counter passes (tools/pmc_all.sh).  Outputs land in gpurun_out/<rNN>_*; tools/round_profiles_collect.py
# copies the summaries into profiles/.
# A third argument selects a part (a gpurun call is limited to 20 minutes): bench | pmc | all (default).
# After a change of the library run `pmc` first, collect (tools/round_profiles_collect.py), THEN `bench` in a call of its own:
# bench.py reads the counter traffic from profiles/*_pmc_traffic*.json and drops a figure recorded for another build.
set -e
R=$1
PART=${2:-all}
export TMPDIR=/tmp
if [ "$PART" != pmc ]; then
timeout -k 10 300 python3 bench.py > gpurun_out/${R}_bench_kyber768.json 2> gpurun_out/${R}_bench_kyber768.err
for w in kyber512 kyber1024 kyber512-90s kyber768-90s kyber1024-90s; do
  timeout -k 10 200 python3 bench.py --workload $w --steps 5 --no-cpu-baseline --no-side-measurements > gpurun_out/${R}_bench_$w.json 2>/dev/null
done
timeout -k 10 200 python3 bench.py --shared-key --steps 5 --no-cpu-baseline > gpurun_out/${R}_bench_kyber768_shared_key.json 2>/dev/null
timeout -k 10 200 python3 bench.py --workload ntt --steps 20 --warmup 3 > gpurun_out/${R}_bench_ntt.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_prof_kem --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-measurements > gpurun_out/${R}_prof_kem.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_prof_ntt --output-format csv -- python3 bench.py --workload ntt --steps 20 --warmup 3 > gpurun_out/${R}_prof_ntt.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_prof_kem90s --output-format csv -- python3 bench.py --workload kyber768-90s --steps 10 --warmup 2 --no-cpu-baseline --no-side-measurements > gpurun_out/${R}_prof_kem90s.log 2>&1
timeout -k 10 300 bash tools/smi_during_bench.sh > gpurun_out/${R}_smi_during_bench.txt 2>&1
fi
if [ "$PART" != bench ]; then
tools/pmc_all.sh $R
# HBM traffic of the 90s kernels (AES sampler and PRF)
export PMC_BENCH_ARGS="--workload kyber768-90s --steps 1 --warmup 1 --batch 524288 --no-cpu-baseline --no-side-measurements"
tools/pmc_pass.sh ${R}_90s_fetch FETCH_SIZE
tools/pmc_pass.sh ${R}_90s_write WRITE_SIZE
fi
