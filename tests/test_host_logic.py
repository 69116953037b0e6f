"""Host-side logic that needs no GPU: the rho histogram, shard arithmetic, bench helpers."""
import numpy as np
import pytest
import torch

from hironaka_amd.rollout import details_from_done_counts, rho_from_details


def reference_details(done_per_step, total, max_length):
    """Literal transcription of the bookkeeping of JAXTrainer.compute_rho's loop (jax_trainer.py:501,519-540)
    for ONE loop: `done_per_step[s]` = number of finished games after s moves, s = 0..max_length-1."""
    details = [0] * max_length
    prev_done, done = 0, done_per_step[0]
    for step in range(max_length - 1):
        details[step] += done - prev_done
        # ... the move happens here ...
        prev_done, done = done, done_per_step[step + 1]
    details[max_length - 1] += total - done
    return details


@pytest.mark.parametrize("seed", range(5))
def test_details_match_reference_loop(seed):
    rng = np.random.default_rng(seed)
    L, total = int(rng.integers(2, 12)), 1000
    # game lengths 0..L+1; those of length exactly L-1 finish on the last move, longer ones never do
    length = rng.integers(0, L + 2, total)
    done = np.array([(length <= s).sum() for s in range(L)])
    assert (length == L - 1).any()
    want = reference_details(list(done), total, L)
    got = details_from_done_counts(torch.as_tensor(done), total)
    assert got == want
    # the reference's quirk: games that finish on the very last move are in no bin
    assert sum(got) == total - int((length == L - 1).sum())
    assert rho_from_details(got) == pytest.approx(sum(want[1:]) / sum(i * n for i, n in enumerate(want)))


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher: N ranks as a CHILD process tree (torch.distributed.run on
    127.0.0.1), the same arguments passed through, the children's status returned -- and nothing of it touches the GPU
    in the parent."""
    import argparse
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 3)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    rc = bench.spawn_ranks(argparse.Namespace(gpus=4))
    cmd = seen["cmd"]
    assert rc == 3 and cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert bench.algorithmic_bytes_per_step(20, 3) == 501 and bench.algorithmic_bytes_per_step(50, 4) == 1625
