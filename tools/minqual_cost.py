"""Kernel time with --min-basequal (the MASK variant: a third column load and the quality compare per step; gapped
records walk their CIGAR) against the default, 2 M records with qualities.  Run on the GPU box."""
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mapdamage_amd import synth  # noqa: E402
from mapdamage_amd.engine import DamageEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    ref = synth.make_genome()
    for name, kw in (("config 2 + qualities", dict(read_len=100)),
                     ("config 3 + qualities", dict(read_len=100, paired=True, frac_softclip=0.10, frac_ins=0.04,
                                                   frac_del=0.04, frac_skip=0.002, frac_hardclip=0.001))):
        b = synth.make_reads(ref, n, 3, contigs=[0, 1], with_qual=True, **kw)
        for q, skew in ((0, False), (20, False), (20, True), (20, "clean"), (20, "clean+hint"), (20, "half clean+hint")):
            import numpy as np
            if skew in ("clean", "clean+hint"):     # no base below the threshold: nothing is masked
                b.qual = np.random.default_rng(9).integers(30, 42, b.qual.shape[0]).astype(np.uint8)
            if skew == "half clean+hint":            # every second record has bases below the threshold
                rng = np.random.default_rng(9)
                q_ = rng.integers(30, 42, b.qual.shape[0]).astype(np.uint8)
                low = (rng.random(b.qual.shape[0]) < 0.05) & (np.repeat(np.arange(b.n) % 2 == 0, np.diff(b.seq_off.astype(np.int64))))
                b.qual = np.where(low, np.uint8(5), q_)
            if isinstance(skew, str) and "hint" in skew:
                from mapdamage_amd.batch import mark_unmaskable
                b.flag = (b.flag & 0x7FFF).astype(np.uint16)
                mark_unmaskable(b, q)
            elif skew is True:   # a sequencer-like distribution: 5 % of the bases below Phred 20, the rest 30..41
                import numpy as np
                rng = np.random.default_rng(9)
                low = rng.random(b.qual.shape[0]) < 0.05
                b.qual = np.where(low, rng.integers(2, 20, b.qual.shape[0]), rng.integers(30, 42, b.qual.shape[0])).astype(np.uint8)
            with DamageEngine([("s", "l")], 70, 10, q, lgd_max=4096) as eng:
                eng.set_reference(ref)
                db = eng.upload(b)
                eng.tabulate(db)
                eng.sync()
                eng.timing(True)
                for _ in range(5):
                    eng.tabulate(db)
                eng.sync()
                n_launch, ms = eng.timing_read()
                db.free()
                print(json.dumps({"workload": name + (", " + skew if isinstance(skew, str) else (", 5 % of the bases below Phred 20" if skew else ", uniform Phred 2..41")),
                                  "min_basequal": q, "kernel_ms": ms / 5,
                                  "Greads_per_s": n / (ms / 5 * 1e-3) / 1e9}), flush=True)


if __name__ == "__main__":
    main()
