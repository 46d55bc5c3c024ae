#!/bin/bash
# the 3 Gb genome (300 copies of the survey's), 25 M config-3 records in random order and sorted: one copy of the 4-bit
# reference against two (MdxTabArgs::ref2)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}
run() {
  tag=$1; shift
  env "$@" python bench.py --reads 25000000 --steps ${STEPS:-20} --warmup 3 --no-cpu --no-secondary --batch-cache /tmp/mdx_bc --tiled-genome 300 $BARGS 2>/dev/null | tail -1 | python -c "
import sys, json
j=json.loads(sys.stdin.readline()); r=j['roofline']; print('%-28s kernel_ms %.4f frac %.4f' % ('$tag', r['kernel_ms'], r['frac']))"
}
run "two copies" MDX_X=0
run "one copy" MDX_NO_REF2=1
BARGS="--sorted" run "two copies, sorted" MDX_X=0
BARGS="--sorted" run "one copy, sorted" MDX_NO_REF2=1
