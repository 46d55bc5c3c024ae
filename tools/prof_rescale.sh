#!/bin/bash
# PMC of rescale_kernel on the rescale benchmark (GPU box, through gpurun)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_rescale
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/tools/rescale_bench.py ${READS:-2000000} $*"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B > $OUT/trace.log 2>&1
for f in $(find $OUT/trace -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats.csv; done
pmc() { local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- $B > $OUT/pmc_$name.log 2>&1
  for f in $(find $OUT/pmc_$name -name '*counter_collection.csv'); do python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f rescale_kernel > $OUT/pmc_$name.txt 2>&1; done; }
pmc inst SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
pmc tcc TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc ta TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
rm -rf $OUT/trace $OUT/pmc_*/
head -4 $OUT/kernel_stats.csv; cat $OUT/pmc_*.txt
