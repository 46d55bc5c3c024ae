#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats, then PMC passes.
# Usage: tools/prof.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu $*"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
for f in $(find $OUT/trace -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats.csv; done
pmc() { # name counters...
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.log 2>&1
  for f in $(find $OUT/pmc_$name -name '*counter_collection.csv'); do
    python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f tabulate_kernel > $OUT/pmc_$name.txt 2>&1
  done
}
pmc inst SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc lat SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES
pmc tcp TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum
pmc tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
pmc write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
rm -rf $OUT/trace $OUT/pmc_*/ 2>/dev/null
ls -la $OUT
