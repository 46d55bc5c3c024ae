#!/bin/bash
# Instruction counters of the tabulation kernel for the in-tree build and prebuilt variants (GPU box).
# usage: tools/pmc_inst.sh tag1 tag2 ...   ("base" = the in-tree build; others = tools/bin/libmdx_<tag>.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/mapdamage_amd/libmdx.so /tmp/libmdx_base.so
for t in "$@"; do
  if [ "$t" = base ]; then cp /tmp/libmdx_base.so $R/mapdamage_amd/libmdx.so; else cp $R/tools/bin/libmdx_$t.so $R/mapdamage_amd/libmdx.so; fi
  touch $R/mapdamage_amd/libmdx.so
  rm -rf /tmp/pmc_$t
  timeout 90 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pmc_$t -o pmc -- python $R/bench.py --steps 5 --warmup 1 --no-cpu $BENCH_ARGS > /tmp/pmc_$t.log 2>&1
  echo "== $t"
  for f in $(find /tmp/pmc_$t -name '*counter_collection.csv'); do python3 $R/tools/pmc_summary.py $f tabulate_kernel; done
done
cp /tmp/libmdx_base.so $R/mapdamage_amd/libmdx.so
