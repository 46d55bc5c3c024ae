#!/bin/bash
# VALU / SALU instructions per record of the tabulation kernel for each variant of tools/split_cost.py (GPU box).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc_split
timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES -d /tmp/pmc_split -o pmc -- python $R/tools/split_cost.py ${1:-2000000} > /tmp/pmc_split.log 2>&1
grep variant /tmp/pmc_split.log
for f in $(find /tmp/pmc_split -name '*counter_collection.csv'); do python3 - $f ${1:-2000000} <<'PY'
import csv, sys
from collections import defaultdict
rows = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "tabulate_kernel" in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
n = float(sys.argv[2])
ids = sorted(rows)
for g in range(0, len(ids), 11):          # 11 launches per variant (1 warm-up + 10 timed)
    c = rows[ids[g + 1]] if g + 1 < len(ids) else rows[ids[g]]
    print("variant %d: VALU/read %.2f SALU/read %.2f LDS/read %.3f VMEM/read %.3f wave-cycles/read %.1f" % (
        g // 11, c.get("SQ_INSTS_VALU", 0) / n, c.get("SQ_INSTS_SALU", 0) / n, c.get("SQ_INSTS_LDS", 0) / n,
        c.get("SQ_INSTS_VMEM_RD", 0) / n, c.get("SQ_WAVE_CYCLES", 0) / n))
PY
done
