"""The N > 1 path end to end on the GPU box: two ranks (gloo; both on the one GPU) run ShardedTriRenderer -- tile-row
bands, gradients written into the flat buffer, ONE all-reduce -- and each compares with the full render."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_match_full_render(hip_device):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(HERE, "sharded_child.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "sharded ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_gpus_2_rehearsal(hip_device):
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself: here both
    share the one GPU over gloo; the line must say n_gpus 2 and carry the sharded parallelism."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DMR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "C2", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["value"] > 0
    assert "x2" in j["config"]["parallelism"]
    assert j["sharded_grad_max_norm_err"] is not None and j["sharded_grad_max_norm_err"] <= 5e-5


def test_bench_emulated_view_band(hip_device):
    """C5 has B = 4 views: rank 5 of 8 renders its (view, band) segments (sharding.view_shares) with B = 1 tensors."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "C5", "--emulate-rank", "5/8", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert "segments [(" in j["config"]["parallelism"] and j["value"] > 0


def test_bench_gpus_2_view_segments(hip_device):
    """`bench.py --gpus 2 --config C5`: two ranks (gloo, one GPU) step their (view, band) segments of the four views with
    B = 1 tensors and join ONE all-reduce over the all-views payload."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DMR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "C5", "--steps", "2",
                        "--warmup", "1", "--settle-ms", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["value"] > 0
    assert "segments of 4 views x2" in j["config"]["parallelism"]
    assert j["config"]["allreduce_payload_bytes"] == 80320640
    assert j["sharded_grad_max_norm_err"] is not None and j["sharded_grad_max_norm_err"] <= 5e-5
