#!/bin/bash
# the packed kernels' launch forms side by side (MdxPkConfig; one build): a block of 1024 with the columns prefetched into
# the LDS, the same without the prefetch, blocks of 512 — the headline launch (25 M config-3 records)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}
run() {
  env "$@" python bench.py --reads 25000000 --steps ${STEPS:-40} --warmup 5 --no-cpu --no-secondary --batch-cache /tmp/mdx_bc $BARGS 2>/dev/null | tail -1 | python -c "
import sys, json
j=json.loads(sys.stdin.readline()); r=j['roofline']; print('%-34s kernel_ms %.4f frac %.4f' % ('$*', r['kernel_ms'], r['frac']))"
}
run MDX_PFL=default
run MDX_PK_THREADS=256
run MDX_PK_THREADS=512
run MDX_PFL=default
