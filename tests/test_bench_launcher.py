"""`python bench.py --gpus N` must start the N ranks itself (the driver's command shape), stay off the GPU in the
parent, relay rank 0's JSON line and exit with the children's return code.  Rehearsed here without a GPU: the
children rendezvous over gloo (MIVIT_BENCH_DRY=1), run one collective and print one line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_self_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"MIVIT_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3.0 and out["steps"] == 3 and out["warmup"] == 1


def test_bench_launcher_propagates_failure():
    # a world size that contradicts --gpus is an error in every rank -> non-zero exit from the parent
    r = _run(["--gpus", "2"], {"MIVIT_BENCH_DRY": "1", "WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0


import pytest


def _bench_json(args, env):
    r = _run(args, env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_over_gloo():
    """The whole multi-rank path of bench.py on the one-GPU box: self-launch, rendezvous, per-stage gradient all-reduce on
    the side stream, max-over-ranks timing -- two ranks on cuda:0, gloo instead of RCCL (RCCL refuses two ranks per device)."""
    out = _bench_json(["--gpus", "2", "--steps", "3", "--warmup", "2", "--batch-per-gpu", "512", "--no-cpu-baseline"],
                      {"MIVIT_BENCH_SHARE_GPU": "1", "MIVIT_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["dist_backend"] == "gloo" and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 1024 and out["value"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_rccl():
    """RCCL (backend "nccl") over xGMI: needs two GPUs, so it runs on the driver's multi-GPU node and is skipped on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    out = _bench_json(["--gpus", "2", "--steps", "3", "--warmup", "2", "--batch-per-gpu", "1024", "--no-cpu-baseline"], {})
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dist_backend"] == "nccl" and out["value"] > 0
