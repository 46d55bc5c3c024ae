#!/bin/bash
# Where the wave cycles of the tabulation kernel go, per record, for the variants of tools/split_cost.py (GPU box):
# parked at s_waitcnt (WAIT_ANY), stalled at issue (WAIT_INST_ANY), issuing (ACTIVE_INST_*), and the memory side.
# usage: tools/pmc_wait.sh [reads] ["variant a|variant b"]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-2000000}
pass() {
  rm -rf /tmp/pmc_w
  MDX_GEN_WORKERS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d /tmp/pmc_w -o pmc -- python $R/tools/split_cost.py $N "${V:-}" > /tmp/pmc_w.log 2>&1
  for f in $(find /tmp/pmc_w -name '*counter_collection.csv'); do python3 - $f $N <<'PY'
import csv, sys
from collections import defaultdict
rows = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "tabulate_kernel" in r["Kernel_Name"]:
        d = rows[int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
n = float(sys.argv[2]); ids = sorted(rows); G = 21
for g in range(0, len(ids), G):
    c = rows[ids[min(g + 1, len(ids) - 1)]]
    print("variant %d: " % (g // G) + " ".join("%s %.2f" % (k.replace("SQ_", ""), v / n) for k, v in sorted(c.items())))
PY
  done
}
V=${2:-}
pass SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
pass SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC
pass TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum
