cd /tmp && export TMPDIR=/tmp
cp $GRAFT_REPO_ROOT/tools/bin/libmdx_p1.so $GRAFT_REPO_ROOT/mapdamage_amd/libmdx.so; touch $GRAFT_REPO_ROOT/mapdamage_amd/libmdx.so
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES -d /tmp/p1 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > /tmp/p1.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/p1 -name '*counter_collection.csv') tabulate_kernel
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1s -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > /tmp/p1s.log 2>&1
grep tabulate $(find /tmp/p1s -name '*kernel_stats.csv')
