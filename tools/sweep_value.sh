#!/bin/bash
# usage: tools/sweep_value.sh tag1 tag2 ... : whole-job value / ms_per_step / kernel_ms of bench.py per prebuilt variant
cd $GRAFT_REPO_ROOT
cp mapdamage_amd/libmdx.so /tmp/libmdx_base.so
for t in "$@"; do
  if [ "$t" = base ]; then cp /tmp/libmdx_base.so mapdamage_amd/libmdx.so; else cp tools/bin/libmdx_$t.so mapdamage_amd/libmdx.so; fi
  touch mapdamage_amd/libmdx.so
  python bench.py --no-cpu 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$t', 'value', round(j['value']/1e9,3), 'ms_per_step', round(j['ms_per_step'],4), 'kernel_ms', round(j['roofline']['kernel_ms'],4))"
done
cp /tmp/libmdx_base.so mapdamage_amd/libmdx.so
