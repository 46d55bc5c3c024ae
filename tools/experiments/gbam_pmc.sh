#!/bin/bash
# Instruction mix and issue activity of the device decode kernels (4 M config-3 records): separate PMC passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pass() {
  rm -rf /tmp/pm; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d /tmp/pm -o pmc -- python $R/tools/gpu_decode_timing.py 4000000 > /tmp/pm.log 2>&1
  for f in $(find /tmp/pm -name '*counter_collection.csv'); do python3 $R/tools/pmc_summary.py $f gbam_inflate_kernel; python3 $R/tools/pmc_summary.py $f gbam_unpack_kernel; done
}
pass SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD
rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/tools/gpu_decode_timing.py 4000000 > /dev/null 2>&1; grep "gbam_\|tabulate" $(find /tmp/tr -name "*kernel_stats.csv") | cut -d, -f1-4
