"""N > 1 path on CPU: two processes, gloo backend, shard-by-read + all-reduce of the tables."""

import os
import pathlib
import subprocess
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_shard_bounds_cover_everything():
    from mapdamage_amd.distributed import shard_bounds
    for n in (0, 1, 7, 1000, 12345):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_pack_unpack_roundtrip():
    from mapdamage_amd import synth
    from mapdamage_amd.tables import pack_words, table_words, unpack_words
    from tests.util import assert_tables_equal, oracle_tableset
    ref, batch = synth.config1_batch()
    libs = [("a", "b"), ("c", "d")]
    ts = oracle_tableset(ref, batch, libs, 70, 10, 0, lgd_max=2048)
    words = pack_words(ts)
    assert words.shape[0] == table_words(2, 70, 10, 2048)
    back = unpack_words(words, libs, 70, 10, 2048, ts.lgd_over)
    assert_tables_equal(back, ts)


def test_two_rank_gloo_allreduce_matches_single_pass():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", str(ROOT / "tests" / "dist_worker.py")]
    out = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "dist ok: world=2" in out.stdout


def test_bench_gpus_n_relaunches_itself_under_torchrun():
    """`python bench.py --gpus 8` must become one rank per GPU (--gpus used to be parsed and ignored)."""
    import json
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "7", "--print-launch"],
                         cwd=str(ROOT), env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")},
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    cmd = json.loads(out.stdout.strip().splitlines()[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "7"] or ("--gpus" in cmd and "--steps" in cmd)
    assert any(part.endswith("bench.py") for part in cmd)


def test_bench_tiled_genome_keeps_the_records_matching():
    """bench.py's 3 Gb-genome workload: the genome repeated as contigs of their own and the records moved to random
    copies (and coordinate-sorted) must count to the same tables as the original records over the original genome —
    only the addresses of the reference windows change."""
    import importlib.util
    from mapdamage_amd import synth
    from tests.util import assert_tables_equal, oracle_tableset
    spec = importlib.util.spec_from_file_location("bench_mod", str(ROOT / "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ref = synth.small_genome()
    batch = synth.make_reads(ref, 4000, 9, len_range=(30, 120), paired=True, frac_softclip=0.1, frac_ins=0.05, frac_del=0.05,
                             frac_skip=0.01)
    libs = [("s", "l")]
    want = oracle_tableset(ref, batch, libs, 70, 10, 0)
    big = bench.tiled_genome(ref, 7)
    assert len(big.names) == 7 * len(ref.names) and sum(big.lengths) == 7 * sum(ref.lengths)
    for srt in (False, True):
        moved = bench.retile(batch, len(ref.names), 7, 123, srt)
        assert moved.n == batch.n and int(moved.tid.max()) >= len(ref.names)
        if srt:
            key = moved.tid.astype(np.int64) * (1 << 32) + moved.pos
            assert (np.diff(key) >= 0).all()
        assert_tables_equal(oracle_tableset(big, moved, libs, 70, 10, 0), want)
    # the traffic look-up never goes silently null
    val, note = bench.traffic_entry("config3", 1000)
    assert val and val > 0 and "per record" in note
    val, note = bench.traffic_entry("no such workload", 1000)
    assert val is None and "no PMC pass committed" in note
