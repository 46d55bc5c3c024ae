#!/bin/bash
# per-kernel durations of tools/rescale_bench.py for prebuilt variants (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/mapdamage_amd/libmdx.so /tmp/libmdx_base.so
for t in "$@"; do
  if [ "$t" = base ]; then cp /tmp/libmdx_base.so $R/mapdamage_amd/libmdx.so; else cp $R/tools/bin/libmdx_$t.so $R/mapdamage_amd/libmdx.so; fi
  rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/tools/rescale_bench.py ${READS:-8000000} ${MODE:-} > /tmp/o.txt 2>&1
  echo "== $t $(tail -1 /tmp/o.txt | grep -o 'bit-exact\|MISMATCH')"; grep "rescale_" $(find /tmp/tr -name "*kernel_stats.csv") | cut -d, -f1-4
done
cp /tmp/libmdx_base.so $R/mapdamage_amd/libmdx.so
