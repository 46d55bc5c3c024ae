#!/bin/bash
# headline launch (25 M config-3 records, 20 timed launches) for several library builds
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}
cp mapdamage_amd/libmdx.so /tmp/libmdx_keep.so
for t in "$@"; do
  if [ "$t" != cur ]; then cp tools/bin/libmdx_$t.so mapdamage_amd/libmdx.so; else cp /tmp/libmdx_keep.so mapdamage_amd/libmdx.so; fi
  touch mapdamage_amd/libmdx.so
  python bench.py --reads 25000000 --steps ${STEPS:-40} --warmup 5 --no-cpu --no-secondary --batch-cache /tmp/mdx_bc $BARGS 2>/dev/null | tail -1 | python -c "
import sys, json
j=json.loads(sys.stdin.readline()); r=j['roofline']; print('%-10s kernel_ms %.4f frac %.4f' % ('$t', r['kernel_ms'], r['frac']))"
done
cp /tmp/libmdx_keep.so mapdamage_amd/libmdx.so
