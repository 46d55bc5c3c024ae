"""N > 1 path on the GPU: shard-by-read over the ranks + RCCL all-reduce of the tables, through
torch.distributed and through the library's own communicator (include/mdx.h mdx_comm_*).  World size 1 runs on any
GPU box; world size 2 needs two devices."""

import os
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


def run_world(n, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "tests" / "gpu_dist_worker.py")]
    out = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-6000:]
    assert "gpu dist ok: world=%d" % n in out.stdout


@pytest.mark.gpu
def test_rccl_world_size_1_reduce_matches_oracle():
    run_world(1, 29551)


@pytest.mark.gpu
def test_rccl_two_ranks_shard_and_reduce_match_single_pass():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    run_world(2, 29552)


@pytest.mark.gpu
def test_two_engines_on_one_gpu_shard_and_reduce_match_oracle():
    """N = 2 on a 1-GPU box: two processes on cuda:0, each a DamageEngine on its shard_bounds slice, the blocks summed
    over gloo; overflow lists and a bad record on rank 1 included (tests/gpu_share_worker.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29554", str(ROOT / "tests" / "gpu_share_worker.py")]
    out = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-6000:]
    assert "gpu share ok: world=2" in out.stdout


@pytest.mark.gpu
def test_bench_line_at_two_ranks_verifies_itself():
    """bench.py --gpus 2 on one GPU (--share-gpu, blocks summed over gloo): the N > 1 line carries `parity` (every
    rank's own block against the oracle) and `reduction` (all-reduced block == sum of the ranks' blocks)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--reads", "300000", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--share-gpu"]
    out = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["parity"].startswith("bit-exact vs oracle (every rank")
    assert line["reduction"].startswith("all-reduced block == sum")
    assert line["roofline"]["frac"] > 0 and "secondary" not in line


@pytest.mark.gpu
def test_bench_line_with_rccl_initialised(tmp_path):
    """bench.py at N = 1 with the process group up: the all-reduce of the timed region goes through RCCL."""
    import json
    env = dict(os.environ, MASTER_PORT="29553", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--reads", "400000", "--steps", "3", "--warmup", "1",
           "--force-dist", "--secondary-reads", "200000", "--genome-copies", "20"]
    out = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["parity"].startswith("bit-exact")
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0
    sec = line["secondary"]
    assert set(sec) == {"config2", "config4", "config5", "minqual", "nlib8", "nlib8_q20", "rescale_file", "file_to_tables", "config3_genome3g", "file_to_tables_50m",
                        "cli_wall"}
    assert sec["minqual"]["q20_kernel_ms"] > 0 and sec["minqual"]["q0_kernel_ms"] > 0
    # (eight libraries: one launch of the packed kernel per call, the resident batch brings its copy ordered by library)
    assert sec["nlib8"]["packed_launches_per_call"] == 1 and sec["nlib8"]["sorts_inside_the_launches"] == 0
    flat = [v for k, v in sec.items() if k not in ("config3_genome3g", "cli_wall")] + list(sec["config3_genome3g"].values())
    assert all(v["parity"].startswith("bit-exact") for v in flat)
    # the command line as a cold process, over the run's own records as a BAM file and over the tiled genome as a FASTA file:
    # the device decode path, the three text files byte-identical to the oracle's tables
    assert set(sec["cli_wall"]) == {"genome10mb_50m", "genome3g_8m"}
    for entry in sec["cli_wall"].values():
        assert entry["wall_s"] > 0 and len(entry["runs"]) == 2
        for run in entry["runs"]:
            assert run["rc"] == 0 and run["decode_path"] == "device" and run["parity"].startswith("misincorporation.txt, dnacomp.txt")
            assert run["stages_s"] and "FASTA file -> resident reference" in run["stages_s"]
    # ... and the line's last key says all of it in a few hundred bytes
    assert list(line)[-1] == "summary" and line["summary"]["parity"] == "bit-exact everywhere"
    assert len(json.dumps(line["summary"])) < 1500 and line["summary"]["cli_wall_s"]["genome10mb_50m"] == sec["cli_wall"]["genome10mb_50m"]["wall_s"]
    assert sec["file_to_tables"]["device_decode"]["reads_per_s"] > 0 and sec["file_to_tables"]["host_decode"]["reads_per_s"] > 0


def _cli(args, gpus=1, port=29557, timeout=900):
    """`python -m mapdamage_amd ...` in a process of its own (the multi-GPU form re-executes itself under torchrun)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "mapdamage_amd"] + [str(a) for a in args]
    if gpus > 1:
        cmd += ["--gpus", str(gpus), "--dist-backend", "gloo", "--share-gpu"]
    out = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-6000:]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("decode", ["--gpu-decode", "--host-decode"])
def test_cli_on_two_ranks_writes_the_reference_tables(tmp_path, decode):
    """`python -m mapdamage_amd --gpus 2` (two ranks on cuda:0, tables summed over gloo: what a 1-GPU box allows): the
    three tables of the reference's golden run, byte for byte, from either decode path."""
    import numpy as np

    from mapdamage_amd import fasta, sam
    from tests.util import Golden
    name = "config1_L70_A10_Q20"
    g = Golden(name)
    rgs = [{"ID": "rg%d" % i, "SM": s, "LB": l} for i, (s, l) in enumerate(g.meta["libraries"])]
    raw_lib = np.load(str(ROOT / "tests" / "golden" / (name + ".npz")))["lib"]
    path = tmp_path / "in.bam"
    sam.write_bam(path, g.batch, g.ref.names, g.ref.lengths, rgs, ["rg%d" % int(l) for l in raw_lib])
    fasta.write_fasta(tmp_path / "ref.fa", g.ref)
    out = tmp_path / "res2"
    _cli(["-i", path, "-r", tmp_path / "ref.fa", "-d", out, "-Q", "20", "--no-stats", decode], gpus=2)
    for f in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt"):
        assert (out / f).read_text() == g.txt[f], f
    log = (out / "Runtime_log.txt").read_text()
    assert "Rank 0 of 2" in log and "Successful run" in log


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [dict(), dict(htslib_blocks=False, block_bytes=65498)], ids=["htslib", "htsjdk"])
def test_cli_on_two_ranks_shards_the_slabs_of_a_file(tmp_path, layout):
    """A file of several slabs: rank r decodes the slabs r, r + 2, ... on the device and steps over the others; the
    tables equal the one-GPU run's byte for byte, and the oracle's.  With records that straddle BGZF blocks and slabs
    too: a slab's records are those that start in it, and a rank that has not seen the slab in front finds its first
    record by the device scan's guess (which the rank that has seen it checks)."""
    import numpy as np

    from mapdamage_amd import fasta, sam, synth
    from oracle import oracle
    ref = synth.make_genome(seed=11, sizes=(("chr1", 300_000), ("chr2", 100_000), ("chrS", 500)), n_run=500, lower_run=3000)
    b = synth.make_reads(ref, 150_000, 21, len_range=(30, 140), paired=True, frac_softclip=0.1, frac_ins=0.04, frac_del=0.04,
                         frac_skip=0.005, frac_filtered=0.03)
    path = tmp_path / "big.bam"
    sam.write_bam(path, b, ref.names, ref.lengths, [{"ID": "rg1", "SM": "s", "LB": "l"}], ["rg1"] * b.n, **layout)
    assert path.stat().st_size > 3 << 20          # three slabs of 1 MiB and more
    fasta.write_fasta(tmp_path / "ref.fa", ref)
    one, two = tmp_path / "one", tmp_path / "two"
    common = ["-i", path, "-r", tmp_path / "ref.fa", "--no-stats", "--chunk-mb", "4", "--log-level", "DEBUG"]
    _cli(common + ["-d", one])
    _cli(common + ["-d", two], gpus=2, port=29558)
    for f in ("misincorporation.txt", "dnacomp.txt", "lgdistribution.txt"):
        assert (one / f).read_bytes() == (two / f).read_bytes(), f
    assert "fallbacks from the device path: 0" in (two / "Runtime_log.txt").read_text()
    # the two ranks share the host: each takes half of the threads the one rank had (mdx_host_threads, sam.usable_cpus)
    import re
    threads = {}
    for name, folder in (("one", one), ("two", two)):
        m = re.search(r"Host threads of this rank: (\d+) inflating beside the device, (\d+) for the host decoder \(LOCAL_WORLD_SIZE (\d+)\)",
                      (folder / "Runtime_log.txt").read_text())
        threads[name] = tuple(int(x) for x in m.groups())
    assert threads["one"][2] == 1 and threads["two"][2] == 2
    assert threads["two"][0] == max(1, threads["one"][0] // 2) and threads["two"][1] == max(1, threads["one"][1] // 2)
    # ... and both are the oracle's
    from mapdamage_amd.tables import TableSet
    w = oracle.tabulate(ref, b, 1, 70, 10)
    want = TableSet([("s", "l")], 70, 10, w["mis"], w["comp"], w["lgd"], w["lgd_over"], w["n_kept"])
    assert (one / "misincorporation.txt").read_text() == want.misincorporation_text()
    assert (one / "dnacomp.txt").read_text() == want.dnacomp_text()
