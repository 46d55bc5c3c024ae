#!/bin/bash
# usage: [READS=8000000] [MODE=plain] tools/sweep_rescale.sh tag1 tag2 ... : kernel_ms of tools/rescale_bench.py per
# prebuilt variant tools/bin/libmdx_<tag>.so ("base" = the in-tree build) on the GPU box
cd $GRAFT_REPO_ROOT
cp mapdamage_amd/libmdx.so /tmp/libmdx_base.so
for t in "$@"; do
  if [ "$t" = base ]; then cp /tmp/libmdx_base.so mapdamage_amd/libmdx.so; else cp tools/bin/libmdx_$t.so mapdamage_amd/libmdx.so; fi
  touch mapdamage_amd/libmdx.so
  python tools/rescale_bench.py ${READS:-2000000} ${MODE:-} 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('rescale', '$t', '${READS:-2000000}', '${MODE:-}', 'kernel_ms', round(j['kernel_ms'],4), j['parity'])"
done
cp /tmp/libmdx_base.so mapdamage_amd/libmdx.so
