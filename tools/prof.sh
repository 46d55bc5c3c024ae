#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats of the bench command (55 dispatches of the
# tabulation kernel: 5 warm-up + 50 timed), then separate PMC passes (8 dispatches each), then the FETCH_SIZE /
# WRITE_SIZE calibration (tools/calib_fetch.hip).  The summary (summary.json) puts the rocprofv3 average of the timed
# dispatches next to the kernel time the same command reports un-profiled (HIP events), and the corrected HBM-side
# bytes per record — the entry that goes into profiles/traffic.json under the workload's key.
# Usage: tools/prof.sh <tag> [bench args, e.g. --config 4 | --tiled-genome 300 [--sorted]]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r03}; shift || true
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# one launch = 25 M records (the bench's resident batches); the generator's forked workers do not run under
# rocprofv3, so a plain run fills the batch cache first
COMMON="--reads 25000000 --no-cpu --no-secondary --batch-cache /tmp/mdx_bc $*"
BENCH="python $R/bench.py --steps 50 --warmup 5 $COMMON"
SHORT="python $R/bench.py --steps 6 --warmup 2 $COMMON"
# (config 5: the rescale kernels behind the fused tabulation kernel belong to the launch)
case "$*" in *"--config 5"*) KPAT="tabulate_kernel|rescale_kernel|rescale_walk_kernel|rescale_reduce_kernel|unpack_listed_kernel";; *) KPAT=tabulate_kernel;; esac
echo "$BENCH" > $OUT/command.txt
$BENCH > $OUT/bench_plain.json 2> $OUT/bench_plain.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
for f in $(find $OUT/trace -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats.csv; done
for f in $(find $OUT/trace -name '*kernel_trace.csv'); do cp $f $OUT/kernel_trace_full.csv; done
pmc() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- $SHORT > $OUT/pmc_$name.log 2>&1
  for f in $(find $OUT/pmc_$name -name '*counter_collection.csv'); do
    python3 $R/tools/pmc_summary.py $f "$KPAT" > $OUT/pmc_$name.txt 2>&1
  done
}
pmc inst SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
if [ -z "${NO_CALIB:-}" ]; then
  # counter calibration on known byte counts beyond the Infinity Cache
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/calib_fetch.hip -o /tmp/calib_fetch 2> $OUT/calib_build.err
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/calib_$c -o pmc -- /tmp/calib_fetch 2 > $OUT/calib_$c.log 2>&1
    for f in $(find $OUT/calib_$c -name '*counter_collection.csv'); do python3 $R/tools/pmc_summary.py $f > $OUT/calib_$c.txt 2>&1; done
  done
fi
python3 $R/tools/prof_summary.py $OUT $TAG > $OUT/summary.json 2> $OUT/summary.err
rm -rf $OUT/trace $OUT/pmc_*/ $OUT/calib_*/ $OUT/kernel_trace_full.csv 2>/dev/null
cat $OUT/summary.json
ls -la $OUT
