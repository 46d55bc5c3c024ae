#!/bin/bash
# usage: tools/sweep_cfg.sh <config> tag1 tag2 ... : kernel_ms of bench.py --config <config> --reads 2000000 per variant
cd $GRAFT_REPO_ROOT
CFG=$1; shift
cp mapdamage_amd/libmdx.so /tmp/libmdx_base.so
for t in "$@"; do
  if [ "$t" = base ]; then cp /tmp/libmdx_base.so mapdamage_amd/libmdx.so; else cp tools/bin/libmdx_$t.so mapdamage_amd/libmdx.so; fi
  touch mapdamage_amd/libmdx.so
  python bench.py --config $CFG --reads 2000000 --no-cpu --steps 10 --warmup 2 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('config $CFG', '$t', 'kernel_ms', round(j['roofline']['kernel_ms'],4))"
done
cp /tmp/libmdx_base.so mapdamage_amd/libmdx.so
