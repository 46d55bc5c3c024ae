#!/bin/bash
# --min-basequal cost (tools/minqual_cost.py, 4 M config-2 records) for several library builds on the GPU box.
# usage: tools/mq_ab.sh tag1 tag2 ...   (tools/bin/libmdx_<tag>.so; "cur" = the in-tree build)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
cp mapdamage_amd/libmdx.so /tmp/libmdx_keep.so
for t in "${@:-cur}"; do
  if [ "$t" != cur ]; then cp tools/bin/libmdx_$t.so mapdamage_amd/libmdx.so; else cp /tmp/libmdx_keep.so mapdamage_amd/libmdx.so; fi
  echo "== $t"; python tools/minqual_cost.py ${READS:-4000000} 2>&1 | grep "config 2" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l); print('%-60s Q%-2d %.4f' % (j['workload'][:60], j['min_basequal'], j['kernel_ms']))"
done
cp /tmp/libmdx_keep.so mapdamage_amd/libmdx.so
