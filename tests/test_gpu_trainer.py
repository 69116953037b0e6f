"""HipTrainer (hironaka_amd/trainer_api.py): the reference's `JAXTrainer.simulate(key, role, use_mcts_policy,
use_unified_tree)` / `compute_rho` entry points over the HIP environment.  Shapes and sanity checks
follow test/testJAXTrainer.py:36-66; BASELINE configs[4] (8192 games x 32 simulations x 20 moves) runs at size."""
import numpy as np
import pytest
import torch

from hironaka_amd import _abi as A
from hironaka_amd import ops
from hironaka_amd.functional import rollout_sanity_tests
from hironaka_amd.rollout import select_sample_after_sim
from hironaka_amd.trainer_api import CONFIG_KEYS, HipTrainer, standin_mlp

pytestmark = pytest.mark.gpu

CONFIG = {  # hironaka/jax/jax_config.yml / test/jax_config.yml, smaller search
    "eval_batch_size": 64, "max_num_points": 20, "dimension": 3, "max_length_game": 20, "max_value": 20,
    "max_grad_norm": 1.0, "scale_observation": True, "reposition": True, "gumbel_scale": 0.3, "use_cuda": True,
    "version_string": "test", "net_type": "dense", "num_evaluations": 10, "num_evaluations_as_opponent": 4,
    "eval_on_cpu": False, "max_num_considered_actions": 10, "discount": 0.99,
}


def make_trainer(**over):
    cfg = dict(CONFIG, **{k: v for k, v in over.items() if k in CONFIG_KEYS})
    m, d = cfg["max_num_points"], cfg["dimension"]
    host_net, host_params = standin_mlp(m * d, 2 ** d - d - 1, 3, width=64)
    agent_net, agent_params = standin_mlp(m * d + d, d, 4, width=64)
    return HipTrainer(42, cfg, host_net=host_net, agent_net=agent_net, host_params=host_params,
                      agent_params=agent_params, use_graph=over.get("use_graph", False),
                      fused_expand=over.get("fused_expand", True))


def legal_successor(obs_t, obs_next, m, d):
    """[B] bool: obs_next equals one HIP step (some host class, some axis) from obs_t -- the consecutive
    observations of a rollout are one legal move apart"""
    b = obs_t.shape[0]
    pts = obs_t[:, : m * d].reshape(b, m, d).contiguous()
    nxt = obs_next[:, : m * d].reshape(b, m, d)
    ok = torch.zeros(b, dtype=torch.bool, device=obs_t.device)
    stages = A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON
    for cls in range(2 ** d - d - 1):
        c = torch.full((b,), cls, dtype=torch.int32, device=obs_t.device)
        for axis in range(d):
            a = torch.full((b,), axis, dtype=torch.int32, device=obs_t.device)
            ok |= (ops.step(pts, c, a, stages=stages)["points"] == nxt).all(dim=2).all(dim=1)
    return ok


def test_config_keys_and_errors():
    with pytest.raises(KeyError):
        HipTrainer(0, {k: v for k, v in CONFIG.items() if k != "discount"}, host_net=lambda f, p: None,
                   agent_net=lambda f, p: None)
    with pytest.raises(TypeError):
        HipTrainer(0, 3)
    with pytest.raises(ValueError):
        HipTrainer(0, CONFIG)
    t = make_trainer()
    assert t.input_dim == {"host": 60, "agent": 63} and t.output_dim == {"host": 4, "agent": 3}
    with pytest.raises(ValueError):
        t.simulate(0, "referee")


@pytest.mark.parametrize("role", ["host", "agent"])
def test_simulate_modes_shapes_and_sanity(role):
    t = make_trainer()
    b, T, m, d = t.eval_batch_size, t.max_length_game, t.max_num_points, t.dimension
    for kw in ({}, {"use_unified_tree": True}, {"use_mcts_policy": True}):
        obs, policy, value = exp = t.simulate(7, role, **kw)
        unified = kw.get("use_unified_tree", False)
        in_dim = (m + 1) * d if (unified or role == "agent") else m * d
        acts = 2 ** d - d - 1 if (unified or role == "host") else d
        assert obs.shape == (b * T, in_dim) and policy.shape == (b * T, acts) and value.shape == (b * T,)
        assert rollout_sanity_tests(exp, (m, d))
        assert torch.isfinite(value).all() and bool((value.abs() <= 1.0 + 1e-6).all())
        if role == "agent" and not unified:
            # every invalid agent action has probability 0 (test/testJAXTrainer.py:62)
            assert bool((obs[:, -d:] - policy >= 0).all())
        mask = select_sample_after_sim(role, exp, d, True, key=5)
        assert mask.shape == (b * T,) and mask.dtype == torch.bool
    # same key, same rollout (Philox-keyed states, deterministic networks, seeded gumbel noise)
    again = t.simulate(7, role, use_mcts_policy=True)
    assert all(torch.equal(torch.nan_to_num(x, neginf=-1e30), torch.nan_to_num(y, neginf=-1e30))
               for x, y in zip(exp, again))


def test_captured_searches_are_replayed_not_recaptured(monkeypatch):
    """simulate(use_mcts_policy=True): the opponent answers with a search of its own, whose argument TUPLES are built
    afresh on every call (functional.mcts_wrapper) around the SAME parameter objects.  The capture cache is keyed by the
    objects: over two simulate calls every (role, tree kind, shape) is captured once and replayed from then on (keyed by
    the tuples' ids it re-captured a search on every opponent move)."""
    from hironaka_amd import search as S
    built = []
    real = S.CapturedSearch

    class Counting(real):
        def __init__(self, *a, **kw):
            built.append(1)
            super().__init__(*a, **kw)

    monkeypatch.setattr(S, "CapturedSearch", Counting)
    import hironaka_amd.simulation_fn as SF
    monkeypatch.setattr(SF, "CapturedSearch", Counting)
    t = make_trainer(use_graph=True)
    first = t.simulate(3, "host", use_mcts_policy=True)
    n_first = len(built)
    again = t.simulate(3, "host", use_mcts_policy=True)
    assert len(built) == n_first, (n_first, len(built))          # the second call replays every graph
    assert n_first <= 4, n_first                                   # ... and the first captured each search once
    assert all(torch.equal(torch.nan_to_num(x, neginf=-1e30), torch.nan_to_num(y, neginf=-1e30))
               for x, y in zip(first, again))


def test_validate_runs_the_reference_schedule():
    """HipTrainer.validate (jax_trainer.py:398-465): seven pairings in the reference's order -- the host network against
    the agent network and the three fixed agents, then the three fixed hosts against the agent network --, each a rho in
    [0, 1] with a histogram of max_length bins; deterministic in the key"""
    t = make_trainer()
    rhos, details = t.validate(batch_size=64, num_of_loops=2, max_length=8, key=5)
    assert len(rhos) == 7 and len(details) == 7
    assert all(len(dt) == 8 and sum(dt) <= 128 for dt in details)
    assert all((r != r) or 0.0 <= r <= 1.0 for r in rhos)
    again = t.validate(batch_size=64, num_of_loops=2, max_length=8, key=5)
    assert again[1] == details


def test_simulate_baseline_config5_at_size():
    """BASELINE configs[4]: batch 8192, 32 simulations per move, 20 moves, dim 3, 20 points, one hipGraph per
    search.  Checked: shapes, rollout_sanity_tests, value range, and that consecutive observations of every game are
    ONE legal environment step apart (12 candidate moves tried per transition with hk_step)."""
    t = make_trainer(eval_batch_size=8192, num_evaluations=32, use_graph=True)
    b, T, m, d = 8192, t.max_length_game, t.max_num_points, t.dimension
    obs, policy, value = exp = t.simulate(11, "host")
    assert obs.shape == (b * T, m * d) and policy.shape == (b * T, 4) and value.shape == (b * T,)
    assert rollout_sanity_tests(exp, (m, d))
    assert bool((value.abs() <= 1.0 + 1e-6).all())
    o = obs.reshape(b, T, m * d)
    for step in (0, 1, 5, T - 2):
        assert bool(legal_successor(o[:, step], o[:, step + 1], m, d).all()), step
    # values of finished games are the discounted ground truth: +0.99^k
    n_pts = (o >= 0).sum(dim=-1) // d
    finished = (n_pts[:, -1] <= 1) & (n_pts[:, 0] > 1)  # (a game that starts finished earns nothing)
    assert finished.any()
    v = value.reshape(b, T)
    assert bool((v[finished] > 0).all())


def test_compute_rho_named_and_network_players():
    import functools
    from hironaka_amd.functional import action_wrapper
    from hironaka_amd.players import choose_first_agent_fn
    t = make_trainer()
    rho, details = t.compute_rho("random", "random", batch_size=512, num_of_loops=2, max_length=12, key=9)
    assert len(details) == 12 and sum(details) <= 1024 and 0 < rho < 1
    # a network host against a fixed agent: the reference-shaped step loop
    host = action_wrapper(functools.partial(t.policy_fns["host"], params=t.host_params), None)
    agent = functools.partial(choose_first_agent_fn, spec=t.spec)
    rho2, det2 = t.compute_rho(host, agent, batch_size=256, num_of_loops=1, max_length=8, key=3)
    assert len(det2) == 8 and sum(det2) <= 256


def test_parameter_objects_do_not_accumulate_captures():
    """a trainer that hands over NEW parameter objects every optimiser step (the functional style of the reference)
    keeps one argument tuple and one captured search per role and kind of tree; in-place updates keep the capture"""
    t = make_trainer(use_graph=True)
    t.simulate(1, "host")
    loop_captures = lambda: sum(len(c.cell_contents) for f in t._sim_fns.values() for c in (f.__closure__ or ())
                                if isinstance(getattr(c, "cell_contents", None), dict))
    first_args = t._fn_args[("host", False)]
    t.simulate(2, "host")
    assert t._fn_args[("host", False)] is first_args  # same objects: same tuples, same capture
    for step in range(3):
        t.host_params = tuple(p.clone() for p in t.host_params)
        t.simulate(3 + step, "host")
    assert len(t._fn_args) == 1 and t._fn_args[("host", False)][0] is t.host_params


def test_expand_operators_against_oracle():
    """the search's expansion operators against oracle/search_oracle.py (index arithmetic in numpy): class ids out of
    range are clamped like hk_decode_host_class; NaN logits win an argmax, the first maximum otherwise"""
    from hironaka_amd._lib import check, lib
    from oracle import search_oracle as SO
    L = lib()
    rng = np.random.default_rng(3)
    dev = lambda a: torch.tensor(a, device="cuda")
    host = lambda t: t.cpu().numpy()
    for (b, n, m, d) in ((1, 2, 4, 3), (37, 5, 20, 3), (300, 9, 10, 4)):
        e, ncls = m * d, 2 ** d - d - 1
        f32 = lambda *shape: rng.standard_normal(shape).astype(np.float32)
        emb, feat = f32(b, n, e), f32(b, n, e)
        parent = rng.integers(0, n, b).astype(np.int32)
        node = rng.integers(0, n, b).astype(np.int32)
        action = rng.integers(-1, ncls + 1, b).astype(np.int32)
        logits, hl = f32(b, d), f32(b, ncls)
        for x in (logits, hl):
            x[rng.random(x.shape) < 0.1] = np.nan
            x[rng.random(x.shape) < 0.1] = 0.5  # ties
        o2, f2 = f32(b, e), f32(b, e)
        # host-role tree
        g_emb, g_feat, g_par, g_act = dev(emb), dev(feat), dev(parent), dev(action)
        obs = torch.empty((b, e), dtype=torch.float32, device="cuda")
        af = torch.empty((b, e + d), dtype=torch.float32, device="cuda")
        check(L.hk_search_expand_gather(g_emb.data_ptr(), g_feat.data_ptr(), g_par.data_ptr(), g_act.data_ptr(),
                                        obs.data_ptr(), af.data_ptr(), b, n, m, d, 0, None), "gather")
        want_obs, want_af = SO.expand_gather(emb, feat, parent, action, d)
        assert np.array_equal(host(obs), want_obs) and np.array_equal(host(af), want_af)
        # the same tables laid out node-major [N, B, E]
        nm_emb, nm_feat = dev(np.ascontiguousarray(emb.transpose(1, 0, 2))), dev(np.ascontiguousarray(feat.transpose(1, 0, 2)))
        obs.zero_(), af.zero_()
        check(L.hk_search_expand_gather(nm_emb.data_ptr(), nm_feat.data_ptr(), g_par.data_ptr(), g_act.data_ptr(),
                                        obs.data_ptr(), af.data_ptr(), b, n, m, d, 1, None), "gather node-major")
        assert np.array_equal(host(obs), want_obs) and np.array_equal(host(af), want_af)
        g_log = dev(logits)
        axis = torch.empty(b, dtype=torch.int32, device="cuda")
        check(L.hk_search_masked_argmax(g_log.data_ptr(), g_act.data_ptr(), axis.data_ptr(), b, d, None), "argmax")
        assert np.array_equal(host(axis), SO.masked_argmax(logits, action, d))
        g_node, g_o2, g_f2 = dev(node), dev(o2), dev(f2)
        check(L.hk_search_expand_scatter(g_o2.data_ptr(), g_f2.data_ptr(), g_node.data_ptr(), g_emb.data_ptr(),
                                         g_feat.data_ptr(), b, n, m, d, None), "scatter")
        want_emb, want_feat = SO.expand_scatter(o2, f2, node, emb, feat)
        assert np.array_equal(host(g_emb), want_emb) and np.array_equal(host(g_feat), want_feat)
        # agent-role tree
        embA, featA = f32(b, n, e + d), f32(b, n, e)
        g_embA, g_featA, g_hl = dev(embA), dev(featA), dev(hl)
        pts = torch.empty((b, e), dtype=torch.float32, device="cuda")
        crd = torch.empty((b, d), dtype=torch.float32, device="cuda")
        check(L.hk_search_expand_gather_agent(g_embA.data_ptr(), g_par.data_ptr(), pts.data_ptr(), crd.data_ptr(),
                                              b, n, m, d, None), "gather_agent")
        want_pts, want_crd = SO.expand_gather_agent(embA, parent, d)
        assert np.array_equal(host(pts), want_pts) and np.array_equal(host(crd), want_crd)
        afeat = torch.empty((b, e + d), dtype=torch.float32, device="cuda")
        cls = torch.empty(b, dtype=torch.int32, device="cuda")
        check(L.hk_search_expand_scatter_agent(g_o2.data_ptr(), g_f2.data_ptr(), g_hl.data_ptr(), g_node.data_ptr(),
                                               g_embA.data_ptr(), g_featA.data_ptr(), afeat.data_ptr(), cls.data_ptr(),
                                               b, n, m, d, ncls, None), "scatter_agent")
        want_embA, want_featA, want_afeat, want_cls = SO.expand_scatter_agent(o2, f2, hl, node, embA, featA, d)
        assert np.array_equal(host(cls), want_cls) and np.array_equal(host(afeat), want_afeat)
        assert np.array_equal(host(g_embA), want_embA) and np.array_equal(host(g_featA), want_featA)
        out = torch.empty_like(g_log)
        check(L.hk_search_mask_logits(g_log.data_ptr(), cls.data_ptr(), out.data_ptr(), b, d, None), "mask_logits")
        assert np.array_equal(host(out), SO.mask_logits(logits, want_cls, d), equal_nan=True)
    assert L.hk_search_expand_gather(None, None, None, None, None, None, 4, 2, 4, 3, 0, None) == A.HK_ERR_NULL
    assert L.hk_search_masked_argmax(None, None, None, 4, 1, None) == A.HK_ERR_SHAPE
    assert L.hk_search_expand_scatter_agent(None, None, None, None, None, None, None, None, 4, 2, 4, 3, 9, None) \
        == A.HK_ERR_SHAPE


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("role", ["host", "agent"])
def test_fused_expansion_equals_generic_path(role, use_graph):
    """simulate(): the expansions through HostExpander / AgentExpander (gather / argmax / step / features / scatter /
    mask operators) give the rollout of the generic recurrent_fn (tensor-library glue) bit for bit"""
    a = make_trainer(eval_batch_size=256, num_evaluations=12, use_graph=use_graph, fused_expand=True)
    b = make_trainer(eval_batch_size=256, num_evaluations=12, use_graph=use_graph, fused_expand=False)
    assert a.fused_expand and not b.fused_expand
    for key in (3, 4):
        x, y = a.simulate(key, role), b.simulate(key, role)
        for u, v in zip(x, y):
            assert torch.equal(torch.nan_to_num(u, neginf=-1e30), torch.nan_to_num(v, neginf=-1e30))
