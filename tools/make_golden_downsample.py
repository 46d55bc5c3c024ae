"""tests/golden/downsample.npz: which records the reference's own BAMReader._downsample_to_fraction and
BAMReader._downsample_to_fixed_number (/root/reference/mapdamage/reader.py:134-164, classmethods over any iterable of
objects with .flag / .reference_id / .reference_start) keep, for seeded flag / tid / pos columns — inputs and outputs only
(build container only: imports the reference with the two stand-in modules of SURVEY 8c).  The draws depend on the flag
column in file order alone (the filter of reader.py:121-132 runs in front); the fixed-number case sorts its reservoir by
(tid, pos) with Python's stable sort, so records with equal coordinates keep their reservoir order — the columns hold
such ties.
    python tools/make_golden_downsample.py
"""
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from tools import ref_harness  # noqa: E402


class Rec:
    __slots__ = ("flag", "reference_id", "reference_start", "index")

    def __init__(self, index, flag, tid, pos):
        self.index, self.flag, self.reference_id, self.reference_start = index, int(flag), int(tid), int(pos)


def main():
    ref_harness.import_reference()
    from mapdamage.reader import BAMReader
    rng = np.random.default_rng(134164)
    n = 5000
    flag = rng.choice(np.array([0, 16, 99, 147, 4, 256, 512, 1024, 2048, 1040], np.uint16), size=n,
                      p=[0.3, 0.3, 0.1, 0.1, 0.04, 0.04, 0.04, 0.04, 0.02, 0.02])
    tid = rng.integers(0, 3, size=n).astype(np.int32)
    pos = rng.integers(0, 400, size=n).astype(np.int32)        # many equal (tid, pos) pairs
    recs = [Rec(i, flag[i], tid[i], pos[i]) for i in range(n)]
    out = {"flag": flag, "tid": tid, "pos": pos}
    cases = []
    for k, (to, seed) in enumerate([(0.3, 7), (0.01, None), (0.999, 123456789), (0.0, 1), (100, 3), (1, 3), (2500, 99), (10_000, 5), (4.7, 11)]):
        if to < 1:
            kept = [r.index for r in BAMReader._downsample_to_fraction(recs, to, seed if seed is not None else 0)]
        else:
            kept = [r.index for r in BAMReader._downsample_to_fixed_number(recs, to, seed)]
        if seed is None:
            seed = 0
        out["kept%d" % k] = np.asarray(kept, np.int64)
        cases.append({"downsample_to": to, "seed": seed, "kept": "kept%d" % k, "n": len(kept)})
        print(cases[-1])
    out["cases"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(ROOT / "tests" / "golden" / "downsample.npz", **out)


if __name__ == "__main__":
    main()
