"""The library's FASTA loader (include/mdx.h mdx_fasta_index / mdx_set_reference_fasta: pysam.FastaFile of
mapdamage/main.py:115 and the fetches of main.py:180, align.py:32-33) and the host's thread budget (mdx_host_threads)."""

import os
import subprocess
import sys

import numpy as np
import pytest

from mapdamage_amd import fasta
from mapdamage_amd.batch import Reference

from util import Golden, assert_tables_equal, oracle_tableset


def _upper_classes(seq: bytes) -> bytes:
    """ref.fetch(...).upper() with everything the loop does not tell apart folded: ACGT, '-', 'N' for the rest."""
    up = np.frombuffer(seq.upper(), np.uint8)
    out = np.full(up.shape, ord("N"), np.uint8)
    for ch in b"ACGT-":
        out[up == ch] = ch
    return out.tobytes()


def test_index_is_built_like_faidx(tmp_path):
    ref = Reference(["a", "b", "c", "e"], [b"ACGT" * 40 + b"AC", b"G" * 61, b"acgtnNRY-" * 7, b"A" * 60])
    path = tmp_path / "t.fa"
    fasta.write_fasta(path, ref, width=60)
    want = (tmp_path / "t.fa.fai").read_text()
    os.remove(tmp_path / "t.fa.fai")
    fasta.ensure_fasta_index(path)
    assert (tmp_path / "t.fa.fai").read_text() == want
    assert fasta.read_fasta_index(str(path) + ".fai") == dict(zip(ref.names, ref.lengths))
    # the name ends at the first white space; CR LF line ends count in the line width, not in the bases
    (tmp_path / "u.fa").write_bytes(b">x desc\r\nACGT\r\nAC\r\n>y\r\nGG\r\n>z\n")
    fasta.ensure_fasta_index(tmp_path / "u.fa")
    assert (tmp_path / "u.fa.fai").read_text() == "x\t6\t9\t4\t6\ny\t2\t23\t2\t4\nz\t0\t30\t0\t0\n"
    # faidx refuses lines of unequal length inside a sequence
    (tmp_path / "v.fa").write_bytes(b">x\nACGT\nAC\nACGT\n")
    with pytest.raises(ValueError, match="different line length"):
        fasta.ensure_fasta_index(tmp_path / "v.fa")
    assert not (tmp_path / "v.fa.fai").exists()
    (tmp_path / "w.fa").write_bytes(b"ACGT\n")
    with pytest.raises(ValueError, match="not a FASTA"):
        fasta.ensure_fasta_index(tmp_path / "w.fa")


def _budget(env):
    code = ("import sys; sys.path.insert(0, %r); from mapdamage_amd.engine import load_library; from mapdamage_amd import sam; "
            "print(load_library().mdx_host_threads(), sam.usable_cpus())" % str(os.path.dirname(os.path.dirname(__file__))))
    full = dict(os.environ)
    for k in ("LOCAL_WORLD_SIZE", "MDX_CPU_MAX_FILE", "MDX_GBAM_HOST_THREADS"):
        full.pop(k, None)
    full.update(env)
    out = subprocess.check_output([sys.executable, "-c", code], env=full).split()
    return int(out[0]), int(out[1])


def test_the_host_is_divided_among_the_ranks_of_a_node(tmp_path):
    """one process per GPU, all of them inflating at once — a rank's pool is its share of what cpu.max grants."""
    cpu_max = tmp_path / "cpu.max"
    cpu_max.write_text("400000 100000\n")          # 4 CPUs' worth of quota
    hw = os.cpu_count() or 1
    half = hw // 2 if hw > 8 else hw
    pool1, cpus1 = _budget({"MDX_CPU_MAX_FILE": str(cpu_max)})
    assert pool1 == max(1, min(half, 4 - 2)) and cpus1 == min(4, len(os.sched_getaffinity(0)))
    pool2, cpus2 = _budget({"MDX_CPU_MAX_FILE": str(cpu_max), "LOCAL_WORLD_SIZE": "2"})
    assert pool2 == max(1, pool1 // 2) and cpus2 == max(1, cpus1 // 2)
    cpu_max.write_text("max 100000\n")
    pool8, cpus8 = _budget({"MDX_CPU_MAX_FILE": str(cpu_max), "LOCAL_WORLD_SIZE": "8"})
    assert pool8 == max(1, half // 8) and cpus8 == max(1, len(os.sched_getaffinity(0)) // 8)
    # a 16-CPU quota shared by the eight ranks of a node: (16 - 2) / 8 threads each, not 14
    cpu_max.write_text("1600000 100000\n")
    pool, _ = _budget({"MDX_CPU_MAX_FILE": str(cpu_max), "LOCAL_WORLD_SIZE": "8"})
    assert pool == max(1, min(half, 14) // 8)
    assert _budget({"MDX_CPU_MAX_FILE": str(cpu_max), "LOCAL_WORLD_SIZE": "8", "MDX_GBAM_HOST_THREADS": "5"})[0] == 5


@pytest.mark.gpu
@pytest.mark.parametrize("width,crlf,piece", [(60, False, 0), (7, False, 4096), (61, True, 4096), (1000, False, 65536)])
def test_fasta_file_becomes_the_resident_reference(tmp_path, monkeypatch, width, crlf, piece):
    from mapdamage_amd import synth
    from mapdamage_amd.engine import DamageEngine
    ref, _ = synth.config1_batch()
    extra = Reference(ref.names + ["odd", "tiny", "none"],
                      ref.seqs + [b"acgtRYKMnN-*xACGT" * 37 + b"A", b"T", b""])
    path = tmp_path / "ref.fa"
    fasta.write_fasta(path, extra, width=width)
    if crlf:
        data = path.read_bytes().replace(b"\n", b"\r\n")
        path.write_bytes(data)
        os.remove(str(path) + ".fai")          # (rebuilt by the library: offsets and widths of the CR LF file)
    if piece:
        monkeypatch.setenv("MDX_FASTA_PIECE_BYTES", str(piece))
    order = ["tiny", ref.names[1], "absent", "odd", ref.names[0], ref.names[2], "none"]
    by_name = dict(zip(extra.names, extra.seqs))
    with DamageEngine([("*", "*")], 70, 10, 0) as eng:
        with pytest.raises(KeyError):
            fasta.reference_for_bam(path, order)
        on_disk = fasta.reference_for_bam(path, order, missing_ok=True)
        eng.set_reference(on_disk)
        assert on_disk.lengths == [len(by_name.get(n, b"")) for n in order]
        for tid, name in enumerate(order):
            seq = by_name.get(name, b"")
            assert eng.reference_fetch(tid, 0, len(seq)) == _upper_classes(seq), name
        n1 = len(by_name[ref.names[1]])
        assert eng.reference_fetch(1, n1 - 5, n1) == _upper_classes(by_name[ref.names[1]][-5:])
        # a name the index lacks, asked for without missing_ok, is the library's error too
        strict = fasta.FastaOnDisk(path, ["absent"], [0], missing_ok=False)
        from mapdamage_amd.engine import MdxError
        with pytest.raises(MdxError, match="not found"):
            eng.set_reference(strict)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["config1_L70_A10_Q0", "edge_L70_A10_Q0", "config1_L70_A10_Q20", "edge_L200_A25_Q10"])
def test_tables_over_the_loaded_fasta_equal_the_reference_golden(tmp_path, name):
    """The goldens' genomes (lowercase stretches, N runs, reads at contig edges) through the file loader."""
    from mapdamage_amd.engine import DamageEngine
    g = Golden(name)
    path = tmp_path / "ref.fa"
    fasta.write_fasta(path, g.ref, width=50)
    with DamageEngine(g.libraries, g.length, g.around, g.minqual) as eng:
        eng.set_reference(fasta.reference_for_bam(path, g.ref.names))
        eng.tabulate(g.batch, packed=True)
        g.check(eng.finish())
