#!/bin/bash
# Instruction counts by part of the packed kernel: tools/pmc_split.sh over the ablation builds abl_NOEVQ (no event is queued),
# abl_NODRAIN (the queue is emptied unread), abl_NOCSA (no carry-save addition), abl_NOFLUSH (the planes are never folded
# into TC), p1 (phase 1 only) — tools/mkvariant.sh abl_X -DMDX_ABL_X; wrong tables, right counts of what is left.
# usage: tools/ablate_pk.sh "variant a|variant b" [reads]
V=$1; N=${2:-4000000}
export MDX_SEQ_4BIT=1
for t in "" abl_NOEVQ abl_NODRAIN abl_NOCSA abl_NOFLUSH abl_NOGRUN p1; do
  echo "== ${t:-cur}"
  LIB=$t $GRAFT_REPO_ROOT/tools/pmc_split.sh $N "$V" | grep "^variant [0-9]*:" | cut -c1-125
done
