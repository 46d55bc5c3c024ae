"""The 4-bit form of the SEQ column (include/mdx.h MDX_SEQ_4BIT): the host packer of the library against a plain numpy
statement of the format — two bases per byte, low nibble first, 1 = A, 2 = C, 4 = T, 8 = G, 0 = anything else (the
reference counts a read symbol only when it is exactly one of "ACGT": mapdamage/statistics.py:27, 101)."""

import numpy as np
import pytest


def numpy_pack(seq):
    lut = np.zeros(256, np.uint8)
    for ch, code in (("A", 1), ("C", 2), ("T", 4), ("G", 8)):
        lut[ord(ch)] = code
    codes = lut[seq]
    if codes.shape[0] & 1:
        codes = np.concatenate([codes, np.zeros(1, np.uint8)])
    return (codes[0::2] | (codes[1::2] << 4)).astype(np.uint8)


@pytest.mark.parametrize("n", [0, 1, 2, 7, 8, 9, 1001, (1 << 21) + 3])
def test_pack_seq_matches_the_format(n):
    from mapdamage_amd.engine import pack_seq
    rng = np.random.default_rng(n)
    # every byte value occurs: lower case, IUPAC letters, '=', 'N', '-', NUL
    seq = rng.choice(np.frombuffer(b"ACGT" * 8 + bytes(range(256)), np.uint8), n).astype(np.uint8)
    want = numpy_pack(seq)
    for threads in (1, 0, 3):
        got = pack_seq(seq, threads=threads)
        assert got.dtype == np.uint8 and got.shape == want.shape
        np.testing.assert_array_equal(got, want)


def test_only_the_four_upper_case_letters_have_a_code():
    from mapdamage_amd.engine import pack_seq
    allb = np.arange(256, dtype=np.uint8)
    packed = pack_seq(allb)
    codes = np.stack([packed & 15, packed >> 4], 1).reshape(-1)
    want = np.zeros(256, np.uint8)
    want[ord("A")], want[ord("C")], want[ord("T")], want[ord("G")] = 1, 2, 4, 8
    np.testing.assert_array_equal(codes, want)
