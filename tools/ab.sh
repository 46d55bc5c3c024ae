#!/bin/bash
# A/B of library builds on the GPU box: for each tag (tools/bin/libmdx_<tag>.so; "cur" = the in-tree build) the
# kernel-only time of the split-cost variants, then (CHECK=1) the fuzz against the oracle.
# usage: [CHECK=1] [VARIANTS="plain paired|config 3"] tools/ab.sh tag1 tag2 ...   -> gpurun_out/ab/<tag>.jsonl
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out/ab
cp mapdamage_amd/libmdx.so /tmp/libmdx_cur.so
for t in "$@"; do
  if [ "$t" = cur ]; then cp /tmp/libmdx_cur.so mapdamage_amd/libmdx.so; else cp tools/bin/libmdx_$t.so mapdamage_amd/libmdx.so; fi
  touch mapdamage_amd/libmdx.so
  python tools/split_cost.py ${READS:-2000000} "${VARIANTS:-}" > gpurun_out/ab/$t.jsonl 2> gpurun_out/ab/$t.err
  python - "$t" <<'PY'
import json,sys
t=sys.argv[1]
rows=[json.loads(l) for l in open('gpurun_out/ab/%s.jsonl'%t) if l.startswith('{')]
print(t, ' | '.join('%s %.4f' % (r['variant'][:14], r['kernel_ms']) for r in rows))
PY
  if [ -n "$CHECK" ]; then
    python tools/fuzz_gpu.py ${FUZZ:-10} > gpurun_out/ab/$t.fuzz 2>&1; echo "$t fuzz: $(tail -1 gpurun_out/ab/$t.fuzz)"
  fi
done
cp /tmp/libmdx_cur.so mapdamage_amd/libmdx.so
