"""The launch forms of the packed kernels that the default run does not take (csrc/mdx_internal.h: MdxPkConfig, MdxTabArgs::ref2),
each in a process of its own — the library reads the knobs once —, against the reference's goldens and the C oracle:

* blocks of 512 and 256 threads (what a long --length falls back to when one block of 1024 with its prefetch areas does not fit the LDS);
* the second copy of the 4-bit reference, 2 GiB + 64 bytes behind the first (a large genome's; forced onto a small one), batches in
  random order — windows read from either copy — and coordinate-sorted — the first copy only."""

import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = textwrap.dedent("""
    import sys
    import numpy as np
    sys.path.insert(0, %r)
    from mapdamage_amd import synth
    from mapdamage_amd.engine import DamageEngine
    from tests.util import Golden, assert_tables_equal, oracle_tableset
    DamageEngine.default_packed = True

    def run(ref, batch, libs, L, A, Q, resident):
        with DamageEngine(libs, L, A, Q, lgd_max=4096) as eng:
            eng.set_reference(ref)
            if resident:
                dev = eng.upload(batch)
                eng.tabulate(dev)
                eng.sync()
                dev.free()
            else:
                eng.tabulate(batch)
            assert eng.packed_launches() == 1
            return eng.finish()

    for name in ("config1_L70_A10_Q0", "config1_L70_A10_Q20", "config3s_L70_A10", "config4s_L70_A10", "edge_L70_A10_Q0", "edge_L8_A3_Q25",
                 "edge_L200_A25_Q10", "fuzz_L70_A10_Q0", "fuzz_L150_A30_Q20", "indelshapes_L70_A10_Q0", "indelshapes_L70_A10_Q20"):
        g = Golden(name)
        g.check(run(g.ref, g.batch, g.libraries, g.length, g.around, g.minqual, False))
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
    cases = [
        (dict(read_len=100, paired=True, frac_softclip=0.10, frac_ins=0.04, frac_del=0.04, frac_skip=0.002, frac_hardclip=0.001), 1, 70, 10, 0, False),
        (dict(read_len=100, paired=True, frac_softclip=0.10, frac_ins=0.04, frac_del=0.04), 1, 70, 10, 0, True),
        (dict(len_range=(35, 150), paired=True, frac_softclip=0.10, frac_ins=0.04, frac_del=0.04, nlib=3, frac_filtered=0.05, frac_n_base=0.02), 3, 70, 10, 0, False),
        (dict(len_range=(20, 160), frac_softclip=0.2, frac_ins=0.1, frac_del=0.1, with_qual=True, nlib=2), 2, 70, 10, 20, False),
        (dict(len_range=(60, 200), frac_softclip=0.1, frac_ins=0.03, frac_del=0.03), 1, 100, 12, 0, False),
    ]
    for k, (kw, nlib, L, A, Q, srt) in enumerate(cases):
        batch = synth.make_reads(ref, 150_000, 40 + k, **kw)
        if srt:
            batch = synth._permute_fixed(batch, np.lexsort((batch.pos, batch.tid)))
        libs = [("S%%d" %% i, "L%%d" %% i) for i in range(nlib)]
        want = oracle_tableset(ref, batch, libs, L, A, Q, lgd_max=4096)
        for resident in (True, False):
            assert_tables_equal(run(ref, batch, libs, L, A, Q, resident), want)
    print("launch form ok")
""" % ROOT)


def _run(env):
    out = subprocess.run([sys.executable, "-c", CODE], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "launch form ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    return out.stderr


@pytest.mark.parametrize("threads", [512, 256])
def test_packed_kernels_in_smaller_blocks(threads):
    err = _run({"MDX_PK_THREADS": str(threads), "MDX_DEBUG_PK": "1"})
    assert "blocks of %d threads" % threads in err, err[-2000:]


def test_the_default_block_is_one_of_1024_threads_per_cu():
    err = _run({"MDX_DEBUG_PK": "1"})
    assert "blocks of 1024 threads" in err and "prefetched into the LDS" in err, err[-2000:]


def test_windows_read_from_the_second_copy_of_the_reference():
    _run({"MDX_REF2_MIN": "0"})
