"""bench.py as the driver runs it: `python bench.py --gpus N` must start its own ranks (child processes) and print
ONE JSON line from rank 0.  The real path (backend nccl = RCCL, one device per rank) needs >= 2 devices; on a
one-GPU box the same flow is rehearsed with ranks sharing the device and the final gather staged through gloo."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(n, extra_env=None, *flags):
    env = dict(os.environ, **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "20", "--warmup", "5",
           "--no-single-step", "--no-cpu-baseline", *flags]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def check_line(out, n):
    assert out["n_gpus"] == n and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "weak"
    assert out["unit"] == "env-steps/s" and out["dtype"] == "f32" and out["value"] > 1e7
    assert out["repeats"] >= 1 and out["timed_steps"] == 20 * out["repeats"] and out["timed_seconds"] >= 0.1
    assert out["games_finished_per_episode"] > 60000
    r = out["roofline"]
    assert r["bound"] == "valu_issue" and (r["frac"] is None or 0 < r["frac"] <= 1)
    assert 0.9 < out["ms_per_step"] / out["ms_per_step_median"] < 1.5
    if n > 1:
        # the trainer-boundary gather follows every region, pipelined behind the next one (`value`); the same regions
        # alone, the gather serialised and timed with its own events, how much of it the pipeline hid, and the check that
        # the pipelined result is the serial gather's
        assert out["gather_us"] > 0 and out["gather_bytes_per_rank"] == n * 65536 * 60 * 4
        assert out["value_without_gather"] >= out["value"] > 0
        assert 0.0 <= out["overlap_hidden_frac"] <= 1.0
        assert "equals the serial gather" in out["gather_check"]


def test_bench_single_rank_line():
    check_line(run_bench(1), 1)


def test_bench_two_ranks_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices (the driver's multi-GPU node); the one-GPU rehearsal runs below")
    check_line(run_bench(2), 2)


def test_bench_two_ranks_rehearsal_on_one_device():
    check_line(run_bench(2, {"HK_BENCH_BACKEND": "gloo"}), 2)
