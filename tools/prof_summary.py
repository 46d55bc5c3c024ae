"""Summary of a tools/prof.sh run: the rocprofv3 kernel trace of the bench command against the kernel time the same
command measures un-profiled, and the HBM-side traffic per record of the profiled workload (the profiles/traffic.json
entry).  usage: prof_summary.py <prof dir> <tag>"""
import csv
import json
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
res = {"tag": tag, "command": open(os.path.join(out, "command.txt")).read().strip()}
line = None
try:
    for ln in open(os.path.join(out, "bench_plain.json")):
        if ln.startswith("{"):
            line = json.loads(ln)
except OSError:
    pass
warm = 5
durs = []
# config 5 (tabulation + rescaling in one call): a launch is the fused tabulation kernel and the rescale kernels behind it
# (rescale_kernel over the records it lists, the walk kernel, the reduction of the summary rows) — their durations summed
config5 = "--config 5" in res["command"]
extra = ("rescale_kernel", "rescale_walk_kernel", "rescale_reduce_kernel", "unpack_listed_kernel") if config5 else ()
try:
    allrows = list(csv.DictReader(open(os.path.join(out, "kernel_trace_full.csv"))))
    rows = [r for r in allrows if "tabulate_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    if extra:
        starts = [int(r["Start_Timestamp"]) for r in rows]
        import bisect
        for r in allrows:
            if any(r["Kernel_Name"].startswith(x) for x in extra):
                j = bisect.bisect_right(starts, int(r["Start_Timestamp"])) - 1     # the tabulation launch in front of it
                if j >= 0:
                    durs[j] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        res["kernels"] = ["tabulate_kernel (rescaling fused in)"] + list(extra)
except (OSError, KeyError):
    pass
if durs:
    timed = durs[warm * (line["config"]["launches_per_step"] if line else 1):]
    res["rocprofv3"] = {"dispatches": len(durs), "avg_ms_all": sum(durs) / len(durs),
                        "timed_dispatches": len(timed), "avg_ms_timed": sum(timed) / max(1, len(timed)),
                        "min_ms": min(durs), "max_ms": max(durs), "first_ms": durs[0]}
if line:
    rl = line["roofline"]
    res["bench_unprofiled"] = {"kernel_ms": rl["kernel_ms"], "launches_timed": rl["launches_timed"], "frac": rl["frac"],
                               "reads_per_launch": rl["reads_per_launch"],
                               "algorithmic_bytes_per_read": rl["algorithmic_bytes_per_read"]}
    if durs:
        res["agreement"] = {"rocprofv3_all_over_bench": res["rocprofv3"]["avg_ms_all"] / rl["kernel_ms"],
                            "rocprofv3_timed_over_bench": res["rocprofv3"]["avg_ms_timed"] / rl["kernel_ms"],
                            "frac_from_rocprofv3_all": rl["algorithmic_bytes_per_read"] * rl["reads_per_launch"] /
                            (res["rocprofv3"]["avg_ms_all"] * 1e-3) / 1e9 / rl["peak"]}


def counter(name, key):
    """Per launch: the tabulation kernel's value per dispatch (config 5: plus the rescale kernels' — each runs once per
    launch, so their per-dispatch values add)."""
    try:
        txt = open(os.path.join(out, "pmc_%s.txt" % name)).read()
    except OSError:
        return None
    tot, found = 0.0, False
    for block in re.split(r"\n(?=\S)", txt):
        head = block.split("\n", 1)[0]
        if "tabulate_kernel" in head or any(head.startswith(x + " ") or head.startswith(x + "(") for x in extra):
            m = re.search(r"%s\s+([0-9.e+]+) per dispatch" % key, block)
            if m:
                tot += float(m.group(1)); found = True
    return tot if found else None


fetch_kb, write_kb = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
if fetch_kb is not None and write_kb is not None and line:
    n = line["roofline"]["reads_per_launch"]
    corr = 1.9      # calibrated (profiles/r02d_calib_*, re-checked by calib_* of this run): this kernel's 12-byte loads
    hbm = fetch_kb * 1024 * corr + write_kb * 1024
    res["traffic"] = {"FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "fetch_correction": corr,
                      "hbm_bytes_per_launch": hbm, "hbm_bytes_per_record": hbm / n,
                      "over_algorithmic": hbm / n / line["roofline"]["algorithmic_bytes_per_read"]}
for nm in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"):
    v = counter("tcc", nm)
    if v is not None:
        res.setdefault("tcc", {})[nm] = v
inst = {k: counter("inst", k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")}
if line and inst["SQ_INSTS_VALU"]:
    n = line["roofline"]["reads_per_launch"]
    res["per_record"] = {k: v / n for k, v in inst.items() if v is not None}
print(json.dumps(res, indent=1))
