"""CPU: `bench.py --gpus N` without a launcher starts N rank processes itself (before touching the GPU), hands each
the environment `torch.distributed.run` would, and reports the first failing rank's status."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_child_environments():
    envs = bench.child_envs(4, 29511, base={"PATH": "/bin", "WORLD_SIZE": "9", "KEEP": "1"})
    assert len(envs) == 4
    for r, e in enumerate(envs):
        assert e["RANK"] == str(r) and e["LOCAL_RANK"] == str(r)
        assert e["WORLD_SIZE"] == "4" and e["LOCAL_WORLD_SIZE"] == "4"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511"
        assert e["KEEP"] == "1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert len({id(e) for e in envs}) == 4


def _run(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=timeout)


def test_flag_and_launcher_must_agree():
    r = _run(["--gpus", "2"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


def test_rccl_run_needs_one_gpu_per_rank():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has the GPUs")
    r = _run(["--gpus", "2"])
    assert r.returncode != 0 and "needs 2 GPUs" in r.stderr


def test_parent_starts_the_ranks_and_reports_their_failure():
    """Without a GPU every rank stops at "needs a HIP device": the parent must come back with a non-zero status, and
    must have got as far as starting ranks without a device of its own."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("meant for the GPU-less container; the GPU tests run the launcher for real")
    r = _run(["--gpus", "2", "--backend", "gloo", "--rays", "4096", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert r.stderr.count("needs a HIP device") == 2, r.stderr


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_print_one_line():
    r = _run(["--gpus", "2", "--backend", "gloo", "--rays", "1048576", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["config"]["rays_total"] == 2 * 1048576
    assert d["config"]["backend"] == "gloo" and d["value"] > 0


@pytest.mark.gpu
def test_two_ranks_spot_workload_reduces_over_the_group():
    r = _run(["--gpus", "2", "--backend", "gloo", "--workload", "spot", "--rays", "262144", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["rays_total"] == 2 * 262144 and "all-reduced" in d["config"]["parallelism"]
