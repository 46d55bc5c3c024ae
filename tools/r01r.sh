cd $GRAFT_REPO_ROOT
bash tools/prof.sh r01r > gpurun_out/prof_r01r.log 2>&1
bash tools/prof_ta.sh r01r >> gpurun_out/prof_r01r.log 2>&1
python bench.py > gpurun_out/r01r_bench.json 2>gpurun_out/r01r_bench.err
python bench.py --config 3 --reads 2000000 --cpu-reads 2000000 > gpurun_out/r01r_bench_config3.json 2>>gpurun_out/r01r_bench.err
python bench.py --config 4 --reads 2000000 --cpu-reads 2000000 > gpurun_out/r01r_bench_config4.json 2>>gpurun_out/r01r_bench.err
python tools/e2e_bench.py 2000000 > gpurun_out/r01r_e2e.json 2>>gpurun_out/r01r_bench.err
python tools/split_cost.py 2000000 > gpurun_out/r01r_split_cost.jsonl 2>>gpurun_out/r01r_bench.err
cat gpurun_out/prof_r01r/kernel_stats.csv | head -5
cat gpurun_out/prof_r01r/pmc_*.txt
for f in gpurun_out/r01r_bench*.json; do python -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'], j.get('parity'), j.get('cpu_baseline',{}).get('value'))"; done
cat gpurun_out/r01r_e2e.json
