"""bench.py at N = 2 in its rehearsal mode (SRH_BENCH_BACKEND=gloo: the two ranks share this box's one GPU, the depth
maps travel through host memory): the code path the driver launches on 2/4/8 GPUs with RCCL -- launcher contract,
sharding, gather, barrier + max-over-ranks timing, ONE JSON line from rank 0 -- for the default pair sharding, the
row-band split of one pair (--shard rows) and the sharded MultiViewStereo run (--workload c4)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _bench2(extra, env_extra=None):
    env = dict(os.environ, SRH_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-rows", "0"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_pairs_and_row_bands():
    a = _bench2(["--workload", "small"])
    assert a["n_gpus"] == 2 and a["steps"] == 2 and a["scaling"] == "weak" and a["value"] > 0
    b = _bench2(["--workload", "small", "--shard", "rows"])
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and b["value"] > 0
    assert "row bands" in b["config"]["parallelism"]


def test_two_rank_multiview():
    c = _bench2(["--workload", "c4"], {"SRH_BENCH_C4_SMALL": "1"})
    assert c["n_gpus"] == 2 and c["scaling"] == "strong" and c["value"] > 0


def test_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2`, plainly: bench.py starts its two ranks itself (child processes, before anything in the
    parent has touched the GPU), rank 0's single JSON line comes through on stdout, the exit status is the ranks'."""
    env = dict(os.environ, SRH_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-rows", "0",
           "--workload", "small"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    a = json.loads(lines[0])
    assert a["n_gpus"] == 2 and a["steps"] == 2 and a["scaling"] == "weak" and a["value"] > 0
    # a failing rank takes the launch down with its exit code instead of leaving the others in a collective
    bad = subprocess.run(cmd, cwd=ROOT, env=dict(env, SRH_LIBRARY="/nonexistent/libstereo_recon_hip.so"),
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]
