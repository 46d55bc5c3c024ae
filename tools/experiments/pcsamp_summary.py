"""Histogram of a rocprofv3 pc_sampling CSV: samples per source line (Instruction_Comment, a -g build) and per instruction."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
print("samples", len(rows), "columns", list(rows[0].keys()) if rows else None)
if not rows:
    sys.exit(0)
keyc = "Instruction_Comment" if "Instruction_Comment" in rows[0] else None
keyi = "Instruction" if "Instruction" in rows[0] else None
for key in (keyc, keyi):
    if not key:
        continue
    h = collections.Counter(r[key] for r in rows)
    print("== by", key)
    for k, v in h.most_common(150):
        print("%7d %5.2f%%  %s" % (v, 100.0 * v / len(rows), k))
for extra in ("Stall_Reason", "Wave_Issued_Instruction", "Instruction_Type", "Stall_Reason_Not_Issued"):
    if extra in rows[0]:
        print("== by", extra)
        for k, v in collections.Counter(r[extra] for r in rows).most_common(30):
            print("%7d %5.2f%%  %s" % (v, 100.0 * v / len(rows), k))
