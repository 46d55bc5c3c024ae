for t in r03d cur; do
  if [ $t != cur ]; then cp mapdamage_amd/libmdx.so /tmp/keep.so; cp tools/bin/libmdx_$t.so mapdamage_amd/libmdx.so; fi
  echo "== $t"; python tools/minqual_cost.py 4000000 2>&1 | grep "config 2" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l); print('%-60s Q%-2d %.4f' % (j['workload'][:60], j['min_basequal'], j['kernel_ms']))"
  if [ $t != cur ]; then cp /tmp/keep.so mapdamage_amd/libmdx.so; fi
done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "basequal or golden or Q" 2>&1 | tail -2
