#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pm; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d /tmp/pm -o pmc -- python $R/tools/gpu_decode_timing.py 4000000 > /tmp/pm.log 2>&1
for f in $(find /tmp/pm -name '*counter_collection.csv'); do python3 $R/tools/pmc_summary.py $f gbam_inflate_kernel; python3 $R/tools/pmc_summary.py $f gbam_unpack_kernel; done
rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/tools/gpu_decode_timing.py 4000000 > /dev/null 2>&1; grep "gbam_\|tabulate" $(find /tmp/tr -name "*kernel_stats.csv") | cut -d, -f1-4
