#!/bin/bash
# Runs bench.py (kernel time only) for each prebuilt variant tools/bin/libmdx_<tag>.so (GPU box).
# usage: tools/sweep_bench.sh tag1 tag2 ...   ("base" = the in-tree build)
cd $GRAFT_REPO_ROOT
cp mapdamage_amd/libmdx.so /tmp/libmdx_base.so
for t in "$@"; do
  if [ "$t" = base ]; then cp /tmp/libmdx_base.so mapdamage_amd/libmdx.so; else cp tools/bin/libmdx_$t.so mapdamage_amd/libmdx.so; fi
  touch mapdamage_amd/libmdx.so
  python bench.py --no-cpu --steps 10 --warmup 2 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    j=json.loads(l); print('$t', 'kernel_ms', round(j['roofline']['kernel_ms'],4), 'frac', round(j['roofline']['frac'],4), j.get('parity'))
except Exception as e:
    print('$t', 'FAILED', l[-300:])
"
done
cp /tmp/libmdx_base.so mapdamage_amd/libmdx.so
