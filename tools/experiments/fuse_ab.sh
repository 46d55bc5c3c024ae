#!/bin/bash
# config-5 kernel times for several library builds (tools/bin/libmdx_<tag>.so; "cur" = the in-tree build)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}
cp mapdamage_amd/libmdx.so /tmp/libmdx_keep.so
for t in "$@"; do
  if [ "$t" != cur ]; then cp tools/bin/libmdx_$t.so mapdamage_amd/libmdx.so; else cp /tmp/libmdx_keep.so mapdamage_amd/libmdx.so; fi
  touch mapdamage_amd/libmdx.so
  python bench.py ${SEQFMT:+--seq-format $SEQFMT} --config 5 --reads ${READS:-25000000} --steps 20 --warmup 5 --no-cpu --no-secondary --batch-cache /tmp/mdx_bc 2>/dev/null | tail -1 | python -c "
import sys, json
j=json.loads(sys.stdin.readline()); r=j['roofline']; print('%-10s total %.3f  tab %.3f  rs %.3f  frac %.4f' % ('$t', r['kernel_ms'], r['tabulate_kernel_ms'], r['rescale_kernel_ms'], r['frac']))"
done
cp /tmp/libmdx_keep.so mapdamage_amd/libmdx.so
