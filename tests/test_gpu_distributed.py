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
    assert set(sec) == {"config2", "config4", "config5", "file_to_tables", "config3_genome3g"}
    flat = [v for k, v in sec.items() if k != "config3_genome3g"] + list(sec["config3_genome3g"].values())
    assert all(v["parity"].startswith("bit-exact") for v in flat)
    assert sec["file_to_tables"]["device_decode"]["reads_per_s"] > 0 and sec["file_to_tables"]["host_decode"]["reads_per_s"] > 0
