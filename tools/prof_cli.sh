#!/bin/bash
# Kernel trace of the command line itself (run on the GPU box through gpurun): an 8 M-record config-3 BAM with qualities is
# written first (forked workers: a process of its own, outside rocprofv3), then `python -m mapdamage_amd -Q 20` runs over it
# under rocprofv3 --kernel-trace --stats — the kernels of the default path from file to tables: inflate, CRC32, scan, unpack
# (which folds the --min-basequal mask into the SEQ column: no pass over the qualities in front of the launches), the packed
# masked kernel.  Usage: tools/prof_cli.sh <tag> [records]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r05_cli}; N=${2:-8000000}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python - <<PY
import sys
sys.path.insert(0, "$R")
import numpy as np
from mapdamage_amd import fasta, sam, synth
ref = synth.make_genome()
b = synth.parallel_batch(dict(read_len=100, paired=True, frac_softclip=0.10, frac_ins=0.04, frac_del=0.04, frac_skip=0.002, frac_hardclip=0.001,
                              contigs=[0, 1], with_qual=True), ref, $N, seed=2020, workers=16)
rng = np.random.default_rng(2020)
low = rng.random(b.qual.shape[0]) < 0.05
b.qual = np.where(low, rng.integers(2, 20, b.qual.shape[0]), rng.integers(30, 42, b.qual.shape[0])).astype(np.uint8)
rgs = [{"ID": "rg%d" % i, "SM": "synthetic", "LB": "lib%d" % (i % 2)} for i in range(4)]
sam.write_bam("/tmp/cli_q.bam", b, ref.names, ref.lengths, rgs, rg_of_record=["rg%d" % (i & 3) for i in range(b.n)], workers=16)
fasta.write_fasta("/tmp/cli_ref.fa", ref)
PY
CMD="python -m mapdamage_amd -i /tmp/cli_q.bam -r /tmp/cli_ref.fa -d /tmp/cli_out --no-stats -Q 20"
echo "$CMD   # $N config-3 records with qualities (5 % below Phred 20), 4 read groups / 2 libraries" > $OUT/command.txt
cd $R
$CMD > $OUT/plain.log 2>&1
( time $CMD ) > $OUT/plain2.log 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- bash -c "cd $R && $CMD" > $OUT/trace.log 2>&1
for f in $(find $OUT/trace -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats.csv; done
rm -rf $OUT/trace
MDX_BAM_TIMING=1 bash -c "cd $R && $CMD" > $OUT/stages.log 2> $OUT/stages.txt
tail -5 $OUT/plain2.log; head -20 $OUT/kernel_stats.csv | cut -c1-150
