"""HIP tree kernels + search driver against oracle/search_oracle.py on the real environment
(parity against the third-party mctx itself is unpinned)."""
import numpy as np
import pytest
import torch

from hironaka_amd import ops
from hironaka_amd.functional import generate_pts, get_reward_fn, make_agent_obs, rollout_sanity_tests
from hironaka_amd.host_action_preprocess import get_batch_decode_from_one_hot
from hironaka_amd.players import choose_first_agent_fn, random_agent_fn, random_host_fn
from hironaka_amd.recurrent_fn import get_recurrent_fn_for_role, get_unified_recurrent_fn
from hironaka_amd.search import RootFnOutput, gumbel_muzero_policy
from hironaka_amd.simulation_fn import get_evaluation_loop, get_simulation
from oracle import search_oracle as SO

pytestmark = pytest.mark.gpu


def host(t):
    return t.detach().cpu().numpy()


def _mlp(in_dim, out_dim, seed):
    """a fixed random two-layer network: (logits, value)"""
    g = torch.Generator().manual_seed(seed)
    w1 = (torch.randn(in_dim, 32, generator=g) / in_dim ** 0.5).cuda()
    w2 = (torch.randn(32, out_dim + 1, generator=g) / 32 ** 0.5).cuda()

    def fn(obs, *args, key=None, **kw):
        h = torch.tanh((obs.clamp(min=-1.0) / 20.0) @ w1) @ w2
        return h[:, :out_dim].contiguous(), torch.tanh(h[:, out_dim]).contiguous()

    return fn


def _as_numpy_recurrent(recurrent_fn):
    def fn(params, key, action, embedding):
        out, emb = recurrent_fn(params, key, torch.as_tensor(action).cuda().long(),
                                torch.as_tensor(embedding).cuda())
        return (host(out.reward), host(out.discount), host(out.prior_logits), host(out.value)), host(emb)
    return fn


@pytest.mark.parametrize("role,spec,num_simulations,max_considered,max_depth",
                         [("host", (20, 3), 32, 16, 20), ("host", (10, 3), 9, 2, 3), ("agent", (20, 3), 16, 3, 16),
                          ("host", (8, 4), 24, 4, 24)])
def test_search_matches_oracle(role, spec, num_simulations, max_considered, max_depth):
    m, d = spec
    b = 96
    pts = generate_pts(7, (b, m, d), 20, torch.float32, False, True)
    ncls = 2 ** d - d - 1
    if role == "host":
        policy_fn = _mlp(m * d, ncls, 3)
        opponent = lambda obs, *a, key=0, **kw: random_agent_fn(obs, spec, key=key)
        root_state = pts.reshape(b, m * d)
        num_actions = ncls
    else:
        policy_fn = _mlp(m * d + d, d, 4)
        opponent = lambda obs, *a, key=0, **kw: random_host_fn(obs.reshape(-1, m, d), key=key)
        coords = get_batch_decode_from_one_hot(d)(random_host_fn(pts, key=5), torch.float32)
        root_state = make_agent_obs(pts, coords)
        num_actions = d
    rf = get_recurrent_fn_for_role(role, policy_fn, opponent, get_reward_fn(role), spec, discount=0.99,
                                   rescale_points=False, reposition=True)
    logits, value = policy_fn(root_state)
    g = torch.Generator().manual_seed(11)
    u = torch.rand((b, num_actions), generator=g).clamp_(1e-20, 1 - 1e-7)
    gumbel = (-torch.log(-torch.log(u)) * 0.3).cuda()
    for invalid in (None, "first"):
        inv = None
        if invalid is not None:
            inv = torch.zeros((b, num_actions), dtype=torch.uint8, device="cuda")
            inv[::2, 0] = 1
        out = gumbel_muzero_policy(((), ()), 123, RootFnOutput(logits, value, root_state), rf, num_simulations,
                                   invalid_actions=inv, max_depth=max_depth,
                                   max_num_considered_actions=max_considered, gumbel=gumbel)
        want = SO.gumbel_muzero_policy(((), ()), host(logits), host(value), host(root_state), _as_numpy_recurrent(rf),
                                       num_simulations, host(gumbel), invalid_actions=None if inv is None else host(inv),
                                       max_depth=max_depth, max_num_considered_actions=max_considered, rng_key=123)
        t, w = out.search_tree, want.search_tree
        for name in ("node_visits", "parents", "action_from_parent", "children_index", "children_visits"):
            assert np.array_equal(host(getattr(t, name)), getattr(w, name)), name
        for name in ("raw_values", "node_values", "children_prior_logits", "children_rewards", "children_discounts",
                     "children_values", "embeddings"):
            assert np.array_equal(host(getattr(t, name)), getattr(w, name)), name
        assert np.array_equal(host(out.action), want.action)
        assert np.allclose(host(out.action_weights), want.action_weights, rtol=0, atol=1e-6)
        assert int(t.children_visits[:, 0].sum(-1).min()) == num_simulations


def test_simulation_shapes_and_sanity():
    """get_evaluation_loop + get_simulation on the host role and on the role-agnostic tree"""
    spec, b, T = (20, 3), 64, 6
    m, d = spec
    ncls = 2 ** d - d - 1
    host_fn = _mlp(m * d, ncls, 1)
    agent_op = lambda obs, *a, key=0, **kw: random_agent_fn(obs, spec, key=key)
    ev = get_evaluation_loop("host", host_fn, agent_op, get_reward_fn("host"), spec, num_evaluations=16, max_depth=10,
                             max_num_considered_actions=4, discount=0.99, rescale_points=False, reposition=True)
    sim = get_simulation("host", ev, b, m, d, T)
    root = generate_pts(3, (b, m, d), 20, torch.float32, False, True).reshape(b, m * d)
    obs, logp, value = sim(42, root)
    assert obs.shape == (b, T, m * d) and logp.shape == (b, T, ncls) and value.shape == (b, T)
    assert torch.equal(obs[:, 0], root)
    p = torch.exp(logp)
    assert torch.allclose(p.sum(-1), torch.ones(b, T, device="cuda"), atol=1e-5)
    assert rollout_sanity_tests((obs.reshape(b * T, -1), logp.reshape(b * T, -1), value.reshape(-1)), spec)
    # consecutive observations are connected by one environment step with SOME host class and axis
    nxt = obs[:, 1].reshape(b, m, d)
    found = torch.zeros(b, dtype=torch.bool, device="cuda")
    for c in range(ncls):
        for ax in range(d):
            cls = torch.full((b,), c, dtype=torch.int32, device="cuda")
            axis = torch.full((b,), ax, dtype=torch.int32, device="cuda")
            stepped = ops.step(obs[:, 0].reshape(b, m, d).contiguous(), cls, axis, stages=7)["points"]
            found |= (stepped == nxt).all(dim=-1).all(dim=-1)
    assert bool(found.all())
    # same seed, same rollouts
    obs2, logp2, value2 = sim(42, root)
    assert torch.equal(obs, obs2) and torch.equal(value, value2)

    # role-agnostic tree: states carry the subset mask tail, both players' moves are searched
    hostu = _mlp(m * d + d, ncls, 5)
    agentu_raw = _mlp(m * d + d, d, 6)

    def agentu(obs, *a, key=None, **kw):
        logits, v = agentu_raw(obs)
        mask = obs[:, m * d:] > 0.5
        return torch.where(mask, logits, torch.full_like(logits, float("-inf"))), v

    evu = get_evaluation_loop("host", hostu, agentu, get_reward_fn("agent"), spec, num_evaluations=12, max_depth=8,
                              max_num_considered_actions=4, discount=0.99, rescale_points=False, reposition=True,
                              role_agnostic=True)
    simu = get_simulation("host", evu, b, m, d, T)
    rootu = torch.cat([root, torch.zeros(b, d, device="cuda")], dim=1)
    obs, logp, value = simu(7, rootu)
    assert obs.shape == (b, T, (m + 1) * d) and logp.shape == (b, T, ncls)
    tails = obs[:, :, m * d:]
    assert bool((tails[:, 0::2].abs().sum(-1) == 0).all())  # host states at even plies
    assert bool((tails[:, 1::2].sum(-1) >= 2).all())        # agent states carry a subset of >= 2 coordinates
    assert rollout_sanity_tests((obs.reshape(b * T, -1), logp.reshape(b * T, -1), value.reshape(-1)), spec)


def test_captured_search_equals_eager():
    """the same search replayed from a hipGraph (CapturedSearch) gives the eager result, call after call"""
    from hironaka_amd.search import CapturedSearch
    spec, b, n = (20, 3), 128, 16
    m, d = spec
    ncls = 2 ** d - d - 1
    policy_fn = _mlp(m * d, ncls, 8)
    opponent = lambda obs, *a, key=0, **kw: choose_first_agent_fn(obs, spec)  # deterministic, capturable
    rf = get_recurrent_fn_for_role("host", policy_fn, opponent, get_reward_fn("host"), spec, discount=0.99,
                                   rescale_points=False, reposition=True)
    cap = None
    for seed in (1, 2, 3):
        root_state = generate_pts(seed, (b, m, d), 20, torch.float32, False, True).reshape(b, m * d)
        logits, value = policy_fn(root_state)
        root = RootFnOutput(logits, value, root_state)
        if cap is None:
            cap = CapturedSearch(((), ()), 5, root, rf, n, max_depth=10, max_num_considered_actions=4, gumbel_scale=0.3)
        got = cap(100 + seed, root)
        want = gumbel_muzero_policy(((), ()), 100 + seed, root, rf, n, max_depth=10, max_num_considered_actions=4,
                                    gumbel_scale=0.3)
        assert torch.equal(got.action, want.action)
        assert torch.equal(got.action_weights, want.action_weights)
        for name in ("node_visits", "children_index", "children_visits", "node_values", "embeddings"):
            assert torch.equal(getattr(got.search_tree, name), getattr(want.search_tree, name)), name
