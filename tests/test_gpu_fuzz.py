"""Randomised parity where the driver sees it: a fixed, seeded slice of tests/fuzz_parity.py and the round-1 fuzzer
mismatch as a regression case.  HIP kernels (through the C ABI) against the C oracle, bit for bit."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

FUZZ_SEED, FUZZ_CASES = 20261004, 600


def test_fuzz_slice():
    """600 seeded random configurations (shapes, semantics, stages, paddings, policies, kernel families): about a
    minute on the GPU box, most of it the single-threaded oracle.  Run once; a failure names its configuration."""
    import fuzz_parity
    assert fuzz_parity.run_cases(FUZZ_CASES, FUZZ_SEED) == FUZZ_CASES


@pytest.mark.parametrize("sem,rstages", [("list", 15), ("list", 7), ("torch", 15), ("jax", 15)])
def test_round1_fuzz_mismatch_regression(sem, rstages):
    """tests/golden/fuzz_regression_20_4.npz: list semantics + rescale on fractional (20,4) states, recording rollout
    of 12 steps -- rows must be ranked BEFORE the rescale rounds their keys together."""
    from hironaka_amd import _abi as A
    from hironaka_amd import ops
    from oracle import c_oracle as CO
    p = np.load(os.path.join(GOLDEN, "fuzz_regression_20_4.npz"))["p"]
    P = torch.as_tensor(p).cuda()
    flags_o = CO.flags_of(sem=sem, noop_if_invalid=sem != "jax", ignore_ended=sem == "torch")
    flags_p = ops.make_flags(sem, sem != "jax", sem == "torch")
    for seed, hp, ap in ((1, A.HK_HOST_RANDOM, A.HK_AGENT_RANDOM), (2, A.HK_HOST_RANDOM, A.HK_AGENT_RANDOM_LEGAL),
                         (3, A.HK_HOST_ALL_COORD, A.HK_AGENT_CHOOSE_LAST), (4, A.HK_HOST_ZEILLINGER, A.HK_AGENT_RANDOM)):
        wp, wrec = CO.rollout(p, 12, seed, host_policy=hp, agent_policy=ap, stages=rstages, flags=flags_o, record=True)
        for force in (0, A.HK_FLAG_FORCE_ONE_LANE, A.HK_FLAG_FORCE_FOUR_LANES, A.HK_FLAG_FORCE_TEAM, A.HK_FLAG_FORCE_GENERIC):
            Q = P.clone()
            rec = ops.rollout(Q, 12, seed, host_policy=hp, agent_policy=ap, stages=rstages, flags=flags_p | force,
                              record=("obs", "host_class", "axis", "done", "reward", "game_length"))
            assert np.array_equal(Q.cpu().numpy(), wp), (sem, rstages, seed, force)
            for k in ("obs", "host_class", "axis", "done", "reward", "game_length"):
                assert np.array_equal(rec[k].cpu().numpy(), wrec[k]), (k, sem, rstages, seed, force)
            Q2 = P.clone()
            ops.rollout(Q2, 12, seed, host_policy=hp, agent_policy=ap, stages=rstages, flags=flags_p | force)
            assert np.array_equal(Q2.cpu().numpy(), wp), ("plain", sem, rstages, seed, force)
