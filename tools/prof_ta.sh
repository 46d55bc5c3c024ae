#!/bin/bash
# Vector-memory pipeline counters (TA / TD / TCP) of tabulate_kernel; run on the GPU box through gpurun.
set -u
TAG=${1:-ta}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu $*"
rocprofv3 -L 2>/dev/null | grep "Counter_Name" | grep -i "TA_\|TD_\|TCP_\|SQ_VMEM\|SQ_INST_LEVEL\|SQ_WAIT" | sort -u > $OUT/counter_names.txt
pmc() { # name counters...
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.log 2>&1
  for f in $(find $OUT/pmc_$name -name '*counter_collection.csv'); do
    python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f tabulate_kernel > $OUT/pmc_$name.txt 2>&1
  done
}
pmc ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pmc ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
pmc ta3 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_max



pmc sqv SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
rm -rf $OUT/pmc_*/ 2>/dev/null
cat $OUT/pmc_*.txt
