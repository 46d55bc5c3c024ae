#!/bin/bash
# the headline launch (25 M config-3 records) of the in-tree build, N times; then config 4 / 5 style workloads with BARGS
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}
for i in $(seq ${N:-2}); do
python bench.py --reads 25000000 --steps ${STEPS:-40} --warmup 5 --no-cpu --no-secondary --batch-cache /tmp/mdx_bc $BARGS 2>/dev/null | tail -1 | python -c "
import sys, json
j=json.loads(sys.stdin.readline()); r=j['roofline']; print('%-10s kernel_ms %.4f frac %.4f' % ('${TAG:-cur}', r['kernel_ms'], r['frac']))"
done
