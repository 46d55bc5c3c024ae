#!/bin/bash
# More PMC passes over the headline command (experiment): instruction cache, fetch levels, memory-instruction latencies,
# FIFO stalls, LDS conflicts.  Usage (gpurun): tools/experiments/pmc_more.sh <tag> [bench args] -> gpurun_out/pmcx_<tag>/
set -u
TAG=${1:-x}; shift || true
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmcx_$TAG
mkdir -p $OUT
COMMON="--reads 25000000 --no-cpu --no-secondary --batch-cache /tmp/mdx_bc $*"
SHORT="python $R/bench.py --steps 6 --warmup 2 $COMMON"
$SHORT > $OUT/plain.json 2> $OUT/plain.err
pmc() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- $SHORT > $OUT/pmc_$name.log 2>&1
  for f in $(find $OUT/pmc_$name -name '*counter_collection.csv'); do python3 $R/tools/pmc_summary.py $f tabulate_kernel > $OUT/pmc_$name.txt 2>&1; done
  rm -rf $OUT/pmc_$name
}
pmc icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES
pmc level SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM
pmc fifo SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES
pmc valu SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
pmc dcache SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ SQC_TC_INST_REQ SQC_TC_STALL SQ_CYCLES SQ_LEVEL_WAVES
cat $OUT/pmc_*.txt
