#!/bin/bash
# inflate stage time of tools/gpu_decode_timing.py for prebuilt variants tools/bin/libmdx_<tag>.so (GPU box)
R=$GRAFT_REPO_ROOT; cd $R
cp mapdamage_amd/libmdx.so /tmp/libmdx_base.so
for t in "$@"; do
  if [ "$t" = base ]; then cp /tmp/libmdx_base.so mapdamage_amd/libmdx.so; else cp tools/bin/libmdx_$t.so mapdamage_amd/libmdx.so; fi
  echo "== $t: $(timeout 600 python tools/gpu_decode_timing.py ${READS:-4000000} 2>&1 | grep 'inflate' | tail -1)"
done
cp /tmp/libmdx_base.so mapdamage_amd/libmdx.so
