"""Constants shared by the host code, the oracle glue and the tests.

The column order follows the reference's ``HEADER`` (mapdamage/seq.py:6-30) with the
derived ``Total`` column removed: the device tables hold the 25 *counted* columns,
``Total`` is recomputed at emit time exactly as mapdamage/statistics.py:192-196 does.

Canonical (output) table layout, all little-endian uint64, C order:

* ``mis [nlib][end][strand][pos < L][col < 25]``
* ``comp[nlib][end][strand][row < L + A][base < 4]``
* ``lgd [nlib][kind][strand][len < lgd_max]``

``end``: 0 = "3p", 1 = "5p" (the sort order of mapdamage/statistics.py:191);
``strand``: 0 = "+", 1 = "-"; ``kind``: 0 = "pe", 1 = "se" (statistics.py:133).
``comp`` rows enumerate the sorted position keys of statistics.py:60-63:
3p: -L..-1, 1..A      5p: -A..-1, 1..L.
"""

LETTERS = ("A", "C", "G", "T")

# 25 counted columns, reference order (mapdamage/seq.py:6-30 minus "Total")
MIS_COLS = (
    "A", "C", "G", "T",
    "G>A", "C>T", "A>G", "T>C", "A>C", "A>T", "C>G", "C>A", "T>G", "T>A", "G>C", "G>T",
    "A>-", "T>-", "C>-", "G>-",
    "->A", "->T", "->C", "->G",
    "S",
)
N_MIS_COLS = len(MIS_COLS)  # 25
COL_S = MIS_COLS.index("S")

# header written to misincorporation.txt (seq.py:30): LETTERS + Total + MUTATIONS
MIS_HEADER = LETTERS + ("Total",) + MIS_COLS[4:]
COMP_HEADER = LETTERS + ("Total",)

ENDS = ("3p", "5p")       # index 0, 1  (sorted order)
STRANDS = ("+", "-")      # index 0, 1
KINDS = ("pe", "se")      # index 0, 1

# symbol classes used on the device and in the C oracle
SYM_A, SYM_C, SYM_G, SYM_T, SYM_GAP, SYM_OTHER = range(6)

# BAM flag bits (mapdamage/reader.py:9-13 + the bits main.py/statistics.py read)
FLAG_PAIRED = 0x1
FLAG_PROPER = 0x2
FLAG_UNMAPPED = 0x4
FLAG_REVERSE = 0x10
FLAG_READ1 = 0x40
FLAG_SECONDARY = 0x100
FLAG_QCFAIL = 0x200
FLAG_DUP = 0x400
FLAG_SUPPLEMENTARY = 0x800
FLAG_FILTER = FLAG_UNMAPPED | FLAG_SECONDARY | FLAG_QCFAIL | FLAG_DUP | FLAG_SUPPLEMENTARY

# BAM CIGAR op codes (mapdamage/align.py:1-11)
OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)
CIGAR_CHARS = "MIDNSHP=X"

QUAL_MISSING = 0xFF  # BAM convention for "no qualities"

# error codes raised by the engine (mirrors include/mdx.h)
MDX_OK = 0
MDX_ERR_ARG = -1
MDX_ERR_HIP = -2
MDX_ERR_STATE = -3
MDX_ERR_MASK_INDEX = -4   # align.py:69-71 IndexError (masked column beyond gapped reference)
MDX_ERR_LGD_OVERFLOW = -5
MDX_ERR_BAD_READ = -6     # tid/pos outside the contig table, CIGAR/SEQ length mismatch
MDX_ERR_COMM = -7         # RCCL failure, or another rank of the communicator reported an error


def comp_positions(end_index, length, around):
    """Sorted position keys for one end (statistics.py:60-63)."""
    if end_index == 0:  # 3p
        return list(range(-length, 0)) + list(range(1, around + 1))
    return list(range(-around, 0)) + list(range(1, length + 1))


def mis_col_index(ref_sym, read_sym):
    """Column of the substitution ``ref>read`` for symbols in A,C,G,T,- ."""
    names = "ACGT-"
    return MIS_COLS.index("%s>%s" % (names[ref_sym], names[read_sym]))
