#!/bin/bash
# Instruction counts by part of the tabulation kernel: tools/pmc_split.sh over ablation builds (tools/bin/libmdx_<tag>.so,
# built with -DMDX_ABL_NO_GRUN / -DMDX_ABL_NO_PRUN / -DMDX_ONLY_PHASE1: wrong tables, right instruction counts of what is left).
# usage: tools/ablate.sh "variant a|variant b" tag1 tag2 ...
V=$1; shift
cp $GRAFT_REPO_ROOT/mapdamage_amd/libmdx.so /tmp/libmdx_keep.so
for t in cur "$@"; do
  echo "== $t"
  if [ $t = cur ]; then cp /tmp/libmdx_keep.so $GRAFT_REPO_ROOT/mapdamage_amd/libmdx.so; LIB= $GRAFT_REPO_ROOT/tools/pmc_split.sh 2000000 "$V" | grep "^variant [0-9]*:" | cut -c1-110
  else LIB=$t $GRAFT_REPO_ROOT/tools/pmc_split.sh 2000000 "$V" | grep "^variant [0-9]*:" | cut -c1-110; fi
done
cp /tmp/libmdx_keep.so $GRAFT_REPO_ROOT/mapdamage_amd/libmdx.so
