#!/bin/bash
# PC sampling of the headline command (experiment; rocprofv3's beta feature): where the tabulation kernel's wavefronts are.
# Usage (through gpurun): tools/experiments/pcsamp.sh <tag> [bench args]  -> gpurun_out/pcsamp_<tag>/
set -u
TAG=${1:-x}; shift || true
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pcsamp_$TAG
mkdir -p $OUT
COMMON="--reads 25000000 --no-cpu --no-secondary --batch-cache /tmp/mdx_bc $*"
python $R/bench.py --steps 2 --warmup 1 $COMMON > $OUT/plain.json 2> $OUT/plain.err
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
for M in stochastic host_trap; do
  if [ $M = stochastic ]; then U=cycles; I=65536; else U=time; I=20; fi
  timeout 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $U --pc-sampling-method $M --pc-sampling-interval $I \
     --kernel-trace --output-format csv -d $OUT/$M -o pcs -- python $R/bench.py --steps 30 --warmup 2 $COMMON > $OUT/$M.log 2>&1
  echo "$M rc=$?" >> $OUT/rc.txt
  for f in $(find $OUT/$M -name '*pc_sampling*.csv'); do
     python3 $R/tools/experiments/pcsamp_summary.py $f > $OUT/${M}_summary.txt 2>&1
     head -5 $f > $OUT/${M}_head.csv
     ls -la $f >> $OUT/rc.txt
  done
  find $OUT/$M -name '*.csv' | head >> $OUT/rc.txt
done
rm -rf $OUT/stochastic $OUT/host_trap
cat $OUT/rc.txt
