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
