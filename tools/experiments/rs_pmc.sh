#!/bin/bash
# VALU/SALU instruction counts and duration of rescale_kernel for prebuilt variants (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/mapdamage_amd/libmdx.so /tmp/libmdx_base.so
for t in "$@"; do
  if [ "$t" = base ]; then cp /tmp/libmdx_base.so $R/mapdamage_amd/libmdx.so; else cp $R/tools/bin/libmdx_$t.so $R/mapdamage_amd/libmdx.so; fi
  rm -rf /tmp/pm; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pm -o pmc -- python $R/tools/rescale_bench.py ${READS:-8000000} ${MODE:-} > /tmp/pm.log 2>&1
  echo "== $t"; for f in $(find /tmp/pm -name '*counter_collection.csv'); do python3 $R/tools/pmc_summary.py $f rescale_kernel; done
  rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/tools/rescale_bench.py ${READS:-8000000} ${MODE:-} > /dev/null 2>&1; grep rescale_kernel $(find /tmp/tr -name "*kernel_stats.csv") | cut -d, -f1-4
done
cp /tmp/libmdx_base.so $R/mapdamage_amd/libmdx.so
