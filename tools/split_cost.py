"""Where the time of survey config 3 goes: kernel-only time of the tabulation pass over 2 M 100 bp records
with one ingredient of the config-3 CIGAR mix at a time.  Run on the GPU box: python tools/split_cost.py [reads]"""
import json
import os
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402

VARIANTS = [
    ("plain paired", {}),
    ("softclip 10%", dict(frac_softclip=0.10)),
    ("ins 4%", dict(frac_ins=0.04)),
    ("del 4%", dict(frac_del=0.04)),
    ("ins 8%", dict(frac_ins=0.08)),
    ("del 8%", dict(frac_del=0.08)),
    ("skip 0.2%", dict(frac_skip=0.002)),
    ("hardclip 0.1%", dict(frac_hardclip=0.001)),
    ("config 3", dict(frac_softclip=0.10, frac_ins=0.04, frac_del=0.04, frac_skip=0.002, frac_hardclip=0.001)),
    ("len 70-150", dict(len_range=(70, 150))),
    ("len 35-150", dict(len_range=(35, 150))),
    ("len 35-69", dict(len_range=(35, 69))),
    ("softclip 30%, 1-40 bases (local alignment)", dict(read_len=150, frac_softclip=0.30, clip_max=40)),
    ("150 bp, no clips", dict(read_len=150)),
    ("config 4", dict(len_range=(35, 150), frac_softclip=0.10, frac_ins=0.04, frac_del=0.04, frac_skip=0.002,
                      frac_hardclip=0.001)),
]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    only = [v for v in sys.argv[2].split("|") if v] if len(sys.argv) > 2 else []
    ref = synth.make_genome()
    # generated before the GPU is touched (forked workers)
    batches = {name: synth.parallel_batch(dict(dict(read_len=100, paired=True, contigs=[0, 1]), **kw), ref, n, 3, workers=int(os.environ.get('MDX_GEN_WORKERS', '64')))
               for name, kw in VARIANTS if not only or name in only}
    with DamageEngine([("s", "l")], 70, 10, 0, lgd_max=4096) as eng:
        eng.set_reference(ref)
        for name, kw in VARIANTS:
            if only and name not in only:
                continue
            b = batches[name]
            db = eng.upload(b)
            eng.tabulate(db)
            eng.sync()
            eng.timing(True)
            for _ in range(20):
                eng.tabulate(db)
            eng.sync()
            n_launch, ms = eng.timing_read()
            eng.timing(False)
            db.free()
            print(json.dumps({"variant": name, "reads": n, "kernel_ms": ms / n_launch,
                              "Greads_per_s": n / (ms / n_launch * 1e-3) / 1e9}), flush=True)


if __name__ == "__main__":
    main()
