"""bench.py --gpus N really starts N ranks (the flag used to be parsed and never read).

CPU only: `--dry-run` stops every rank after the rendezvous (gloo), before anything touches a GPU; the `-m gpu`
counterpart (tests/test_sharded_gpu.py::test_bench_gpus_2_rehearsal) runs the real bench with two gloo ranks on one GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _run(args, env=None, timeout=300):
    return subprocess.run([sys.executable, BENCH] + args, env=env or _clean_env(), capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_flag_starts_n_ranks(n):
    r = _run(["--gpus", str(n), "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["world_size_env"] == n and j["gpus_flag"] == n


def test_single_rank_has_no_children():
    r = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_under_a_launcher_the_flag_must_match_world_size():
    env = dict(_clean_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = _run(["--gpus", "4", "--dry-run"], env=env)
    assert r.returncode != 0 and "does not match" in r.stderr


def test_a_failing_rank_fails_the_launch(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\ntime.sleep(60)\n")
    rc = bench.launch_ranks(2, [], script=str(child), timeout_s=30)
    assert rc == 3


def test_too_few_devices_is_an_error():
    # this container has no GPU: with the nccl backend --gpus 2 must refuse, not run one rank and call it two
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices present")
    env = dict(_clean_env(), DMR_DIST_BACKEND="nccl")
    r = _run(["--gpus", "2"], env=env)
    assert r.returncode != 0 and "HIP devices" in r.stderr
