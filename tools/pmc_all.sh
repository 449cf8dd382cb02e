#!/bin/bash
# tools/pmc_all.sh <prefix> -- the round's counter passes: SQ issue/wait counters, LDS, and the two
# traffic passes (FETCH_SIZE and WRITE_SIZE cannot share a pass), for the KEM pipeline and for the
# polynomial kernels (config 2).  Each pass is its own short run.
set -e
P=$1
tools/pmc_pass.sh ${P}_sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
tools/pmc_pass.sh ${P}_sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
tools/pmc_pass.sh ${P}_fetch FETCH_SIZE
tools/pmc_pass.sh ${P}_write WRITE_SIZE
export PMC_BENCH_ARGS="--workload ntt --steps 3 --warmup 1"
tools/pmc_pass.sh ${P}_ntt_sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
tools/pmc_pass.sh ${P}_ntt_sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
tools/pmc_pass.sh ${P}_ntt_fetch FETCH_SIZE
tools/pmc_pass.sh ${P}_ntt_write WRITE_SIZE
