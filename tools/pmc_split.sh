#!/bin/bash
# VALU / SALU / LDS / VMEM instructions and wave cycles per record of the tabulation kernel for the variants of
# tools/split_cost.py (GPU box).  usage: tools/pmc_split.sh [reads] ["variant a|variant b"]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ -n "$LIB" ]; then cp $R/tools/bin/libmdx_$LIB.so $R/mapdamage_amd/libmdx.so; touch $R/mapdamage_amd/libmdx.so; fi
rm -rf /tmp/pmc_split
MDX_GEN_WORKERS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d /tmp/pmc_split -o pmc -- python $R/tools/split_cost.py ${1:-2000000} "${2:-}" > /tmp/pmc_split.log 2>&1
grep variant /tmp/pmc_split.log | python3 -c "
import sys, json
for i, l in enumerate(sys.stdin):
    j = json.loads(l); print('variant %d = %s: %.4f ms' % (i, j['variant'], j['kernel_ms']))"
for f in $(find /tmp/pmc_split -name '*counter_collection.csv'); do python3 - $f ${1:-2000000} <<'PY'
import csv, sys
from collections import defaultdict
rows = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "tabulate_kernel" in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
n = float(sys.argv[2])
ids = sorted(rows)
G = 21                                     # launches per variant (1 warm-up + 20 timed)
for g in range(0, len(ids), G):
    c = rows[ids[g + 1]] if g + 1 < len(ids) else rows[ids[g]]
    print("variant %d: VALU/read %.2f SALU/read %.2f LDS/read %.3f VMEM/read %.3f wave-cycles/read %.1f wait-any/read %.1f active-valu/read %.1f busy-cycles %.0f" % (
        g // G, c.get("SQ_INSTS_VALU", 0) / n, c.get("SQ_INSTS_SALU", 0) / n, c.get("SQ_INSTS_LDS", 0) / n,
        c.get("SQ_INSTS_VMEM_RD", 0) / n, c.get("SQ_WAVE_CYCLES", 0) / n, c.get("SQ_WAIT_INST_ANY", 0) / n,
        c.get("SQ_ACTIVE_INST_VALU", 0) / n, c.get("SQ_BUSY_CYCLES", 0)))
PY
done
