#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}
cp mapdamage_amd/libmdx.so /tmp/libmdx_keep.so
for t in "$@"; do
  if [ "$t" != cur ]; then cp tools/bin/libmdx_$t.so mapdamage_amd/libmdx.so; else cp /tmp/libmdx_keep.so mapdamage_amd/libmdx.so; fi
  touch mapdamage_amd/libmdx.so
  echo "== $t"; python tools/experiments/wave_clk.py 25000000 2>&1 | grep -E "total duration|block end|block index"
done
cp /tmp/libmdx_keep.so mapdamage_amd/libmdx.so
