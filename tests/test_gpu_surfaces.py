"""GPU tests of the reference-named python surfaces (functional API, codec, players, recurrent_fn,
HipPoints, gym environments, rollouts) -- they read like the reference's own tests
(test/testJAX.py, test/testTensorPoints.py, test/testGymEnv.py, test/testGame.py) and check against
the same known answers, the committed outputs of the reference's siblings, and the oracle."""
import numpy as np
import pytest
import torch

from hironaka_amd import _abi as A
from hironaka_amd import ops
from hironaka_amd.agent import Agent, ChooseFirstAgent, RandomAgent
from hironaka_amd.core import HipPoints
from hironaka_amd.fused_game import FusedGame
from hironaka_amd.game import GameHironaka
from hironaka_amd.functional import (flatten, generate_pts, get_done_from_flatten, get_dones, get_feature_fn,
                                     get_preprocess_fns, get_reward_fn, get_take_actions, make_agent_obs)
from hironaka_amd.gym_env import HironakaAgentEnv, HironakaHostEnv
from hironaka_amd.host import AllCoordHost, RandomHost, Zeillinger
from hironaka_amd.host_action_preprocess import (HostActionEncoder, batch_encode, batch_encode_one_hot, decode_table,
                                                 get_batch_decode, get_batch_decode_from_one_hot)
from hironaka_amd.players import (all_coord_host_fn, choose_first_agent_fn, choose_last_agent_fn,
                                  get_host_with_flattened_obs, random_agent_fn, random_host_fn, zeillinger_fn,
                                  zeillinger_fn_slice)
from hironaka_amd.recurrent_fn import get_recurrent_fn_for_role, get_unified_recurrent_fn
from hironaka_amd.rollout import (compute_rho, details_from_done_counts, rho_from_details, rollout_postprocess,
                                  select_sample_after_sim, simulate_fixed_policies)
from oracle import c_oracle as CO
from oracle import np_oracle as NO

pytestmark = pytest.mark.gpu


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x)).to(dtype).cuda()


def host(t):
    return t.detach().cpu().numpy()


# ---- test/testJAX.py:80-153, 205-219, 461-488 -------------------------------------------------------

def test_functional_known_answers(golden):
    e = golden["take_actions_inputs"]
    p, c, a = dev(e["points"]), dev(e["coords"]), dev(e["axis"])
    take = get_take_actions("host", (4, 3), rescale_points=True, reposition=False)
    out = take(flatten(p), c, a)
    want = ops.rescale(ops.get_newton_polytope(ops.shift(p, c, a)))
    assert out.shape == (2, 12) and torch.equal(out, flatten(want))
    combined = make_agent_obs(p, c)
    assert combined.shape == (2, 15)
    take = get_take_actions("agent", (4, 3), rescale_points=False, reposition=False)
    out = take(combined, a, a)
    assert torch.equal(out, flatten(ops.get_newton_polytope(ops.shift(p, c, a))))
    with pytest.raises(ValueError):
        get_take_actions("referee", (4, 3))
    with pytest.raises(ValueError):
        get_reward_fn("referee")
    with pytest.raises(ValueError):
        get_preprocess_fns("referee", (4, 3))
    obs_pre, coords_pre = get_preprocess_fns("agent", (4, 3))
    assert torch.equal(obs_pre(combined), p) and torch.equal(coords_pre(combined, None), c)
    e = golden["reward_convention"]
    done, prev = torch.tensor(e["done"]).cuda(), torch.tensor(e["prev_done"]).cuda()
    assert host(get_reward_fn("host")(done, prev)).tolist() == e["host"]
    assert np.array_equal(host(get_reward_fn("agent")(done, prev)), np.array(e["agent"], dtype=np.float32))
    assert host(get_done_from_flatten(combined, "agent", 3)).tolist() == host(get_dones(p)).tolist()


def test_feature_fn(golden):
    e = golden["features_host"]
    assert np.array_equal(host(get_feature_fn("host", (6, 3))(dev(e["points"]))), np.array(e["expected"], dtype=np.float32))
    for key in ("features_agent_unscaled", "features_agent_scaled"):
        e = golden[key]
        fn = get_feature_fn("agent", tuple(e["spec"]), scale_observation=e["scale_observation"])
        assert np.array_equal(host(fn(dev(e["obs"]))), np.array(e["expected"], dtype=np.float32)), key


def test_codec(golden, live_torch):
    e = golden["codec_d3"]
    assert np.array_equal(host(decode_table(3)), np.array(e["decode_table"]))
    enc_in = dev(e["encode_in"], torch.int32)
    assert host(batch_encode(enc_in)).tolist() == e["encode_out"]
    one_hot = batch_encode_one_hot(enc_in)
    assert np.array_equal(host(one_hot), np.eye(4, dtype=np.float32)[e["encode_out"]])
    assert torch.equal(get_batch_decode_from_one_hot(3)(one_hot), enc_in)
    assert torch.equal(get_batch_decode(3)(dev(e["encode_out"], torch.int32)), enc_in)
    for d in range(2, 8):
        tab = decode_table(d)
        assert np.array_equal(host(tab), live_torch[f"codec/d{d}"])
        assert host(batch_encode(tab)).tolist() == list(range(2 ** d - d - 1))
    with pytest.raises(ValueError):
        get_batch_decode(11)


# ---- test/testJAX.py:232-276 ----------------------------------------------------------------------

def test_players(golden):
    e = golden["all_coord_host"]
    obs = dev(e["points"])
    assert host(all_coord_host_fn(obs)).tolist() == [[0, 0, 0, 1], [0, 0, 0, 1]]
    rh = random_host_fn(obs, key=3)
    assert rh.shape == (2, 4) and bool((rh.sum(1) == 1).all())
    e = golden["zeillinger_slice"]
    assert host(zeillinger_fn_slice(dev(e["points"])[0])).tolist() == [0, 1, 0, 0]
    e = golden["zeillinger_batch"]
    want = batch_encode_one_hot(dev(e["expected_mask"], torch.int32))
    assert torch.equal(zeillinger_fn(dev(e["points"])), want)
    e = golden["zeillinger_padded"]
    assert host(zeillinger_fn(dev(e["points"]))).tolist() == [[0, 1, 0, 0]]
    flat_host = get_host_with_flattened_obs((10, 3), zeillinger_fn)
    assert host(flat_host(dev(e["points"]).reshape(1, 30))).tolist() == [[0, 1, 0, 0]]
    e = golden["choose_first_last"]
    agent_obs = make_agent_obs(dev(golden["all_coord_host"]["points"]), dev(e["coords"]))
    assert host(choose_first_agent_fn(agent_obs, (2, 3))).argmax(1).tolist() == e["first"]
    assert host(choose_last_agent_fn(agent_obs, (2, 3))).argmax(1).tolist() == e["last"]
    ra = random_agent_fn(agent_obs, (2, 3), key=1)
    assert ra.shape == (2, 3) and bool((ra.sum(1) == 1).all())


def test_recurrent_fn_matches_oracle_composition():
    """recurrent_fn.py:84-121: one expansion of each role == the oracle's step with the same actions."""
    spec, b = (10, 3), 64
    m, d = spec
    pts = generate_pts(5, (b, m, d), 20, torch.float32, False, True)
    uniform = lambda x, *a, **k: (torch.zeros((x.shape[0], 4 if x.shape[1] == m * d else d), device=x.device),
                                  torch.zeros(x.shape[0], device=x.device))
    # host role: our host plays class ids, the fixed agent answers choose-first
    rf = get_recurrent_fn_for_role("host", uniform, lambda o, **k: choose_first_agent_fn(o, spec),
                                   get_reward_fn("host"), spec, reposition=True)
    cls = torch.randint(0, 4, (b,), device="cuda", dtype=torch.int32)
    out, nxt = rf(((), ()), 0, cls, flatten(pts))
    mask = NO.decode_class(host(cls), d)
    axis = mask.argmax(1).astype(np.int32)
    want = CO.step(host(pts), host(cls), axis, stages=7)
    assert np.array_equal(host(nxt).reshape(b, m, d), want["points"])
    assert np.array_equal(host(out.reward), want["reward"])
    assert out.prior_logits.shape == (b, 4) and float(out.discount[0]) == pytest.approx(0.99)
    # agent role: our agent plays an axis on an agent observation, the fixed host answers all-coord
    coords = ops.decode_host_class(cls, d, torch.float32)
    agent_obs = make_agent_obs(pts, coords)
    rf = get_recurrent_fn_for_role("agent", uniform, lambda o, **k: all_coord_host_fn(o.reshape(-1, m, d)),
                                   get_reward_fn("agent"), spec, reposition=True)
    ax = torch.randint(0, d, (b,), device="cuda", dtype=torch.int32)
    out, nxt = rf(((), ()), 0, ax, agent_obs)
    want = CO.step(host(pts), host(cls), host(ax), stages=7, reward_sign=-1.0)
    assert nxt.shape == (b, m * d + d)
    assert np.array_equal(host(nxt[:, : m * d]).reshape(b, m, d), want["points"])
    assert bool((nxt[:, m * d:] == 1).all()) and np.array_equal(host(out.reward), want["reward"])
    # unified tree: host node -> agent node -> host node
    host_fn = lambda x, *a, **k: (torch.zeros((x.shape[0], 4), device=x.device), torch.zeros(x.shape[0], device=x.device))
    agent_fn = lambda x, *a, **k: (torch.zeros((x.shape[0], d), device=x.device), torch.zeros(x.shape[0], device=x.device))
    uf = get_unified_recurrent_fn(host_fn, agent_fn, get_reward_fn("host"), spec, reposition=True)
    state0 = torch.cat([flatten(pts), torch.zeros((b, d), device="cuda")], dim=1)
    out1, state1 = uf(((), ()), 0, cls, state0)
    assert torch.equal(state1[:, m * d:], coords) and torch.equal(state1[:, : m * d], flatten(pts))
    assert bool(torch.isinf(out1.prior_logits[:, d:]).all()) and float(out1.discount[0]) == pytest.approx(-0.99)
    out2, state2 = uf(((), ()), 0, ax, state1)
    assert bool((state2[:, m * d:] == 0).all())
    assert np.array_equal(host(state2[:, : m * d]).reshape(b, m, d), CO.step(host(pts), host(cls), host(ax), stages=7)["points"])


# ---- test/testTensorPoints.py:48-198 ----------------------------------------------------------------

def test_hip_points_reference_sequence(golden):
    pts = HipPoints(dev(golden["newton_r"]["points"]))
    assert pts.batch_size == 2 and pts.max_num_points == 4 and pts.dimension == 4
    pts.get_newton_polytope()
    assert np.array_equal(host(pts.points), np.array(golden["newton_r"]["expected"], dtype=np.float32))
    q = pts.shift([[1, 2], [0, 2, 3]], [1, 3], inplace=False)
    assert q is not pts and np.array_equal(host(pts.points), np.array(golden["newton_r"]["expected"], dtype=np.float32))
    pts.shift([[1, 2], [0, 2, 3]], [1, 3])
    pts.get_newton_polytope()
    assert np.array_equal(host(pts.points), np.array(golden["shift_then_newton_r2"]["expected"], dtype=np.float32))
    assert torch.equal(q.get_newton_polytope().points, pts.points)
    pts.reposition()
    assert np.array_equal(host(pts.points), np.array(golden["reposition_r3"]["expected"], dtype=np.float32))
    pts.rescale()
    assert np.allclose(host(pts.points), np.array(golden["rescale_rs"]["expected"], dtype=np.float32), atol=1e-6)
    assert hash(pts) == hash(pts.copy()) and not pts.ended
    assert host(pts.get_num_points()).tolist() == [3, 2]
    # invalid actions / finished games are no-ops (test/testTensorPoints.py:84-103)
    e = golden["torch_invalid_actions"]
    pts = HipPoints(dev(e["points"]))
    pts.shift([[1], [0, 2, 3]], [0, 1])
    assert np.array_equal(host(pts.points), np.array(e["expected"], dtype=np.float32))
    e = golden["torch_ended_game"]
    pts = HipPoints(dev(e["points"]))
    pts.shift([[0, 1]], [1], ignore_ended_games=True)
    assert np.array_equal(host(pts.points), np.array(e["expected_ignore_ended"], dtype=np.float32)) and pts.ended
    pts.shift([[0, 1]], [1], ignore_ended_games=False)
    assert np.array_equal(host(pts.points), np.array(e["expected_forced"], dtype=np.float32))
    e = golden["torch_functions_2"]
    pts = HipPoints(dev(e["points"]))
    pts.get_newton_polytope()
    assert np.array_equal(host(pts.points), np.array(e["expected"], dtype=np.float32))
    e = golden["torch_rescale_by_0"]
    pts = HipPoints(dev(e["points"]))
    pts.rescale()
    assert bool(pts.points.isfinite().all())
    pts.type(torch.float16)
    assert pts.dtype == torch.float16 and pts.points.dtype == torch.float16
    pts.get_newton_polytope()
    assert pts.points.dtype == torch.float16
    ragged = HipPoints([[[1, 2, 3], [2, 3, 4]], [[1, 1, 1], [4, 4, 4], [9, 8, 7]]], max_num_points=4)
    assert ragged.points.shape == (2, 4, 3) and host(ragged.get_num_points()).tolist() == [2, 3]
    with pytest.raises(TypeError):
        HipPoints(torch.zeros(1, 2, 3), device="cpu")


def test_hip_points_fused_game_move(live_torch):
    """FusedGame.agent_move (trainer/fused_game.py:150-163) on HipPoints == the reference's TensorPoints run,
    as three launches and as the fused HipPoints.step."""
    for tag in ("m20_d3_f32_pad1", "m10_d3_f64_padeps", "m50_d4_f32_pad1"):
        g = lambda k: live_torch[f"{tag}/{k}"]
        pad = -1.0 if tag.endswith("pad1") else -1e-8
        tdt = torch.float64 if "f64" in tag else torch.float32
        pts = HipPoints(torch.as_tensor(g("points")), padding_value=pad, dtype=tdt)
        pts.get_newton_polytope()
        assert np.array_equal(host(pts.points), g("game_start"))
        fused = pts.copy()
        mask, axis = torch.as_tensor(g("mask")).cuda(), torch.as_tensor(g("axis")).cuda()
        pts.shift(mask.type(tdt), axis.type(tdt))
        pts.get_newton_polytope()
        assert np.array_equal(host(pts.points), g("game_unscaled"))
        assert np.array_equal(host(pts.ended_batch_in_tensor), g("game_ended"))
        pts.rescale()
        assert np.array_equal(host(pts.points), g("game_scaled"))
        res = fused.step(mask, axis, rescale=True, want=("done",))
        assert np.array_equal(host(fused.points), g("game_scaled"))
        assert np.array_equal(host(res["done"]), g("game_ended"))
        feats = pts.get_features()
        assert bool((feats[:, :-1, 0] >= feats[:, 1:, 0]).all())


# ---- gym surface (test/testGymEnv.py, test/testGame.py) -----------------------------------------------

def test_host_env_scaled_matches_oracle_replay():
    """scale_observation=True: per step shift+newton, THEN the host chooses on the unscaled state, THEN
    rescale (hironaka_host_env.py:46-68) -- replayed with the oracle in exactly that order."""
    n, m, d = 48, 10, 3
    env = HironakaHostEnv(Zeillinger(), dimension=d, max_num_points=m, max_value=20, num_envs=n, seed=5)
    obs = env.reset()
    raw = NO.random_ints(n, m, d, 20, 5).astype(np.float64)
    state = CO.rescale(CO.get_newton_polytope(raw, sem="list"), sem="list")
    state = CO.get_newton_polytope(state, sem="list")

    def choose(unscaled):
        cls = CO.zeillinger(unscaled, sem="list")
        return np.where((cls >= 0)[:, None], NO.decode_class(np.maximum(cls, 0), d), 0)

    coords = choose(state)
    state = CO.rescale(state, sem="list")
    for t in range(12):
        assert np.array_equal(host(obs["points"]), state.astype(np.float32)), t
        assert np.array_equal(host(obs["coords"]), coords.astype(np.float64)), t
        axis = np.where(coords.sum(1) > 0, coords.argmax(1), 0).astype(np.int32)
        obs, reward, stopped, _ = env.step(torch.as_tensor(axis).cuda())
        legal = coords.sum(1) > 0
        unscaled = CO.step(state, coords.astype(np.float64), axis, stages=A.HK_STAGE_SHIFT | A.HK_STAGE_NEWTON,
                           flags=CO.flags_of(sem="list", noop_if_invalid=True))
        ended = unscaled["num_points"] <= 1
        assert np.array_equal(host(stopped), ended)
        assert np.array_equal(host(reward), np.where(legal, (~ended).astype(np.float64), -1e-3))
        coords = np.where(ended[:, None], 0, choose(unscaled["points"]))
        state = CO.rescale(unscaled["points"], sem="list")
    assert bool(stopped.any())


@pytest.mark.parametrize("scale", [0])
def test_host_env_replays_reference_game(live_list, scale):
    """The reference's GameHironaka trajectory (Zeillinger host vs recorded agent axes, config 1) through
    the vectorised HironakaHostEnv: same states, same host choices, reward/stop conventions.  (Only the
    unscaled run: with scale_observation the game object lets the host choose on the rescaled state while
    the gym env chooses before rescaling -- see test_host_env_scaled_matches_oracle_replay.)"""
    states = live_list[f"game_scale{scale}/states"]
    masks, axes = live_list[f"game_scale{scale}/masks"], live_list[f"game_scale{scale}/axes"]
    env = HironakaHostEnv(Zeillinger(), dimension=3, max_num_points=10, max_value=20, num_envs=32,
                          scale_observation=bool(scale))
    obs = env.reset(points=live_list[f"game_scale{scale}/start"])
    assert env.current_step == 1  # the reference's post-reset pseudo step (hironaka_host_env.py:38-39)
    assert np.array_equal(host(obs["points"]), states[0].astype(np.float32))
    for t in range(len(masks)):
        alive = states[t][:, 1, 0] >= 0
        assert np.array_equal(host(obs["coords"])[alive], masks[t][alive].astype(np.float64)), t
        act = torch.as_tensor(np.where(axes[t] >= 0, axes[t], 0)).cuda()
        obs, reward, stopped, info = env.step(act)
        assert np.array_equal(host(obs["points"]), states[t + 1].astype(np.float32)), t
        ended = states[t + 1][:, 1, 0] < 0
        legal = alive & (axes[t] >= 0)
        assert np.array_equal(host(stopped), ended)
        assert np.array_equal(host(reward)[legal], (~ended[legal]).astype(np.float64))
        assert (host(reward)[~legal] == -1e-3).all()
    assert bool(stopped.all()) and (host(obs["coords"]) == 0).all()


def test_host_env_single_game_interface():
    env = HironakaHostEnv(Zeillinger(), dimension=3, max_num_points=10, max_value=10, seed=3)
    obs = env.reset()
    assert isinstance(obs["points"], np.ndarray) and obs["points"].shape == (10, 3) and obs["coords"].shape == (3,)
    assert obs["points"].dtype == np.float32 and obs["points"].max() <= 1.0
    steps = 0
    done = False
    while not done and steps < 200:
        action = int(np.flatnonzero(obs["coords"])[0]) if obs["coords"].sum() else 0
        obs, reward, done, info = env.step(action)
        assert isinstance(reward, float) and isinstance(done, bool)
        steps += 1
    assert done and info["current_step"] == steps + 1
    o2, r2, d2, _ = env.step(0)  # stepping a stopped env: illegal move (no coordinates offered)
    assert r2 == pytest.approx(-1e-3) and d2


def test_agent_env_matches_oracle():
    """HironakaAgentEnv with ChooseFirstAgent: every step == the oracle's list-mode step with axis = lowest
    chosen coordinate; reward +1 exactly when the game ends; threshold penalties."""
    n, m, d = 64, 10, 3
    env = HironakaAgentEnv(ChooseFirstAgent(), dimension=d, max_num_points=m, max_value=12, num_envs=n,
                           scale_observation=False, step_threshold=6, seed=11)
    obs = env.reset()
    state = host(env._points)
    assert np.array_equal(state, CO.get_newton_polytope(NO.random_ints(n, m, d, 12, 11).astype(np.float64), sem="list"))
    rng = np.random.default_rng(0)
    for t in range(6):
        cls = rng.integers(0, 4, n).astype(np.int32)
        mask = NO.decode_class(cls, d)
        obs, reward, stopped, info = env.step(torch.as_tensor(mask).cuda())
        axis = mask.argmax(1).astype(np.int32)
        want = CO.step(state, mask.astype(np.float64), axis, stages=A.HK_STAGE_SHIFT | A.HK_STAGE_NEWTON,
                       flags=CO.flags_of(sem="list", noop_if_invalid=True))
        state = want["points"]
        assert np.array_equal(host(env._points), state), t
        assert np.array_equal(host(obs), state.astype(np.float32))
        ended = want["num_points"] <= 1
        trip = (t + 1) >= 6
        assert np.array_equal(host(reward), ended.astype(np.float64) - (6.0 if trip else 0.0))
        assert np.array_equal(host(stopped), ended | trip)
        assert host(info["last_action_taken"]).tolist() == axis.tolist()
    # discrete host actions are decoded as the RAW binary expansion (reference quirk)
    env2 = HironakaAgentEnv(RandomAgent(seed=1), use_discrete_actions_for_host=True, num_envs=4, dimension=3)
    env2.reset()
    env2.step(torch.tensor([3, 5, 6, 7]).cuda())
    assert bool((env2.last_action_taken >= 0).all())
    env2.step(torch.tensor([1, 2, 4, 0]).cuda())  # fewer than two coordinates: the agent returns None
    assert bool((env2.last_action_taken == -1).all())


def test_hosts():
    pts = ops.generate_points(128, 10, 3, 20, seed=1, dtype=torch.float64)
    assert bool((AllCoordHost().select_coord(pts).sum(1)[ops.get_num_points(pts) >= 2] == 3).all())
    rnd = RandomHost(seed=5).select_coord(pts)
    alive = ops.get_num_points(pts) >= 2
    assert bool((rnd.sum(1)[alive] == 2).all()) and bool((rnd.sum(1)[~alive] == 0).all())
    z = Zeillinger().select_coord(pts)
    want = CO.zeillinger(host(pts), sem="list")
    assert np.array_equal(host(z), np.where((want >= 0)[:, None], NO.decode_class(np.maximum(want, 0), 3), 0))
    assert Zeillinger.get_char_vector((5, 0, -3)) == (8, 2)  # test/testZeillinger.py:15-19


# ---- rollouts ------------------------------------------------------------------------------------------

def test_compute_rho_fused_equals_stepwise():
    """compute_rho (jax_trainer.py:467-556): the fused kernel and the reference-shaped step-by-step loop give
    the same histogram for deterministic policies, and the fused path equals the oracle for random ones."""
    kw = dict(spec=(20, 3), batch_size=2048, max_value=20, max_length=12, num_of_loops=2, key=9)
    rho_f, det_f = compute_rho("all_coord", "choose_first", **kw)
    rho_s, det_s = compute_rho(lambda o, **k: all_coord_host_fn(o.reshape(-1, 20, 3)),
                               lambda o, **k: choose_first_agent_fn(o, (20, 3)), **kw)
    assert det_f == det_s and rho_f == rho_s and sum(det_f) <= 4096  # last-move finishers are in no bin
    rho, det = compute_rho("random", "random", **kw)
    want = np.zeros(12, dtype=np.int64)
    for loop in range(2):
        p0 = CO.generate_points(2048, 20, 3, 20, 9 + loop)
        _, rec = CO.rollout(p0, 11, 9 + loop, record=False)
        want += rec["done_count"].astype(np.int64)
    assert det == details_from_done_counts(torch.as_tensor(want), 4096)
    assert rho == pytest.approx(rho_from_details(det)) and 0 < rho < 1


@pytest.mark.parametrize("loops", [1, 2, 3, 5])
def test_compute_rho_one_launch_equals_the_oracle(loops):
    """compute_rho with named policies is ONE launch (hk_rollout_desc.gen_max_value + episodes: the batches are drawn
    inside the kernel, no state is stored): the histogram equals the sum of the oracle's per-loop histograms of
    generate_points(key + loop) -> rollout(key + loop), for 1 / 2 / an odd number of loops, twice (cached workspace)"""
    kw = dict(spec=(20, 3), batch_size=1536, max_value=20, max_length=10, num_of_loops=loops, key=21)
    for _ in range(2):
        rho, det = compute_rho("random", "random_legal", **kw)
        want = np.zeros(10, dtype=np.int64)
        for loop in range(loops):
            p0 = CO.generate_points(1536, 20, 3, 20, 21 + loop)
            _, rec = CO.rollout(p0, 9, 21 + loop, agent_policy=A.HK_AGENT_RANDOM_LEGAL, record=False)
            want += rec["done_count"].astype(np.int64)
        assert det == details_from_done_counts(torch.as_tensor(want), 1536 * loops)
        assert rho == pytest.approx(rho_from_details(det))


@pytest.mark.parametrize("spec,host_name,dtype", [((50, 4), "zeillinger", torch.float32), ((20, 3), "zeillinger", torch.float32),
                                                   ((7, 3), "random", torch.float32), ((10, 3), "random", torch.float64),
                                                   ((20, 4), "all_coord", torch.float32)])
def test_compute_rho_every_route_equals_the_oracle(spec, host_name, dtype):
    """compute_rho by name on the fused kernel ((50,4) incl. Zeillinger's host, (20,4)) and on the library's generate +
    rollout composition (Zeillinger's host on a small shape, a shape without a four-lane kernel, float64): the same
    histogram as the oracle's generated rollouts"""
    m, d = spec
    hp = {"random": A.HK_HOST_RANDOM, "zeillinger": A.HK_HOST_ZEILLINGER, "all_coord": A.HK_HOST_ALL_COORD}[host_name]
    rho, det = compute_rho(host_name, "random", spec=spec, batch_size=700, max_value=20, max_length=9, num_of_loops=3,
                           key=5, dtype=dtype)
    _, want = CO.rollout_generated(700, spec, 8, 5, max_value=20, episodes=3, host_policy=hp,
                                   dtype=np.float32 if dtype == torch.float32 else np.float64)
    assert det == details_from_done_counts(torch.as_tensor(want["done_count"].astype(np.int64)), 2100)


def test_simulate_shapes_and_values():
    spec, b, T = (20, 3), 512, 20
    for role, obs_dim, act_dim in (("host", 60, 4), ("agent", 63, 3)):
        obs, policy, value = simulate_fixed_policies(3, role, spec=spec, batch_size=b, max_value=20, max_length_game=T)
        assert obs.shape == (b * T, obs_dim) and policy.shape == (b * T, act_dim) and value.shape == (b * T,)
        assert bool((policy.sum(1) == 1).all())
        sign = 1.0 if role == "host" else -1.0
        assert bool(((value * sign) >= 0).all()) and bool(((value * sign) <= 1.0).all())
    # the first observation of every game is its initial state
    p0 = CO.generate_points(b, 20, 3, 20, 3)
    obs, _, value = simulate_fixed_policies(3, "host", spec=spec, batch_size=b, max_value=20, max_length_game=T)
    assert np.array_equal(host(obs).reshape(b, T, 60)[:, 0], p0.reshape(b, 60))
    _, rec = CO.rollout(p0, T, 3, record=False)
    gl = rec["game_length"]
    v0 = host(value).reshape(b, T)[:, 0]
    fin = gl > 0
    assert np.allclose(v0[fin], 0.99 ** (gl[fin] - 1), rtol=1e-5)
    assert np.all(v0[gl == 0] == 0)  # done at entry: no reward, no estimate (jax/util.py:261-284)
    # every value equals the oracle's restatement of rollout_postprocess on the same observations
    for role in ("host", "agent"):
        obs, _, value = simulate_fixed_policies(3, role, spec=spec, batch_size=b, max_value=20, max_length_game=T)
        want = NO.rollout_postprocess(host(obs).reshape(b, T, -1), 3, 0.99, role, use_unified_tree=False)
        assert np.allclose(host(value), want, rtol=0, atol=1e-6)


def test_rollout_postprocess_reference_vectors_on_device():
    """test/testJAXTrainer.py:91-389 through the product's device-side rollout_postprocess."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "rollout_postprocess.json")) as f:
        doc = json.load(f)
    for case in doc["cases"]:
        obs = dev(case["obs"])
        b, t, _ = obs.shape
        rollouts = (obs, torch.zeros(b, t, 4, device="cuda"), torch.zeros(b, t, device="cuda"))
        _, _, v = rollout_postprocess(rollouts, case["role"], doc["dimension"], doc["discount"], case["unified"])
        assert np.allclose(host(v), np.asarray(case["expected"], np.float32).ravel(), rtol=0, atol=1e-6)


def test_select_sample_after_sim():
    """jax/util.py:351-382: unfinished states are always kept; with mixing, exactly as many extra draws as
    there are unfinished states (so at most twice their number in total)"""
    spec, b, T = (20, 3), 256, 20
    for role, extra in (("host", 0), ("agent", 3)):
        obs, policy, value = simulate_fixed_policies(5, role, spec=spec, batch_size=b, max_value=20, max_length_game=T)
        undone = (obs >= 0).sum(-1) > 3 + extra
        plain = select_sample_after_sim(role, (obs, policy, value), 3, mix_random_terminal_states=False)
        assert torch.equal(plain, undone)
        mixed = select_sample_after_sim(role, (obs, policy, value), 3, key=9)
        n = int(undone.sum())
        assert bool((mixed | ~undone).all()) and n <= int(mixed.sum()) <= 2 * n
        assert torch.equal(mixed, select_sample_after_sim(role, (obs, policy, value), 3, key=9))
        assert 0 < n < b * T


# ---- src/_fn.py:241-325, game.py:84-119, trainer/fused_game.py:54-163 ---------------------------------

def test_host_action_encoder(live_torch):
    """test/testUtil.py:226-236 + the reference's own tables"""
    enc = HostActionEncoder(3)
    assert [enc.decode(i) for i in range(4)] == [[0, 1], [0, 2], [1, 2], [0, 1, 2]]
    for d in range(2, 8):
        enc = HostActionEncoder(d)
        n = 2 ** d - d - 1
        ids = torch.arange(n, dtype=torch.int32).cuda()
        table = enc.decode_tensor(ids)
        assert table.dtype == torch.float32 and np.array_equal(host(table), live_torch[f"codec/d{d}"])
        assert np.array_equal(host(enc.encode_tensor(table)), live_torch[f"codec/d{d}_roundtrip"])
        for i in range(n):
            coords = enc.decode(i)
            assert enc.encode(coords) == i and host(table[i]).nonzero()[0].tolist() == coords


class _ReplayAgent(Agent):
    """plays the axes the reference's RandomAgent drew when the fixture was recorded"""

    def __init__(self, axes):
        self.axes, self.t = axes, 0

    def _get_actions(self, points, coords):
        self.t += 1
        return torch.as_tensor(self.axes[self.t - 1]).to(points.device)


@pytest.mark.parametrize("scale", [0, 1])
def test_game_hironaka_replays_reference_game(live_list, scale):
    """BASELINE config 1 (dim 3, 10 points, 32 games, Zeillinger vs the recorded agent): GameHironaka over a
    list-semantics HipPoints reproduces the reference's game -- every state, every host choice, the stop"""
    g = lambda k: live_list[f"game_scale{scale}/{k}"]
    states, masks, axes = g("states"), g("masks"), g("axes")
    pts = HipPoints(torch.as_tensor(g("start")), dtype=torch.float64, semantics="list", value_threshold=1e8)
    game = GameHironaka(pts, Zeillinger(), _ReplayAgent(axes), scale_observation=bool(scale))
    assert np.array_equal(host(game.state.points), states[0]) and not game.stopped
    for t in range(len(masks)):
        went_on = game.step()
        assert np.array_equal(host(game.coord_history[-1]), masks[t]), t
        assert np.array_equal(host(game.move_history[-1]), axes[t]), t
        assert np.array_equal(host(game.state.points), states[t + 1]), t
        assert went_on == (t + 1 < len(masks))
    assert game.stopped and game.state.ended and not game.step()
    assert len(game.coord_history) == len(masks)


class _SetNet(torch.nn.Module):
    """the players of tests/golden/make_golden.py::make_fused_game, rebuilt from the committed arrays"""

    def __init__(self, w, bias, v=None):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor(w), requires_grad=False)
        self.bias = torch.nn.Parameter(torch.tensor(bias), requires_grad=False)
        self.v = None if v is None else torch.nn.Parameter(torch.tensor(v), requires_grad=False)

    def forward(self, x):
        if isinstance(x, dict):
            return x["points"].sum(dim=1) @ self.w + x["coords"] @ self.v + self.bias
        return x.sum(dim=1) @ self.w + self.bias


def _canonical(rows):
    """[n, m, d]: every game's rows in one canonical order (the reference leaves ties in coordinate 0 open)"""
    out = np.empty_like(rows)
    for i, g in enumerate(rows):
        out[i] = g[np.lexsort(g.T)[::-1]]
    return out


@pytest.mark.parametrize("role", ["host", "agent"])
@pytest.mark.parametrize("spec", [(20, 3), (8, 4)])
def test_fused_game_step_matches_reference(live_fused, spec, role):
    """FusedGame.step for three moves == the reference's FusedGame over TensorPoints with the same players:
    experiences of the unfinished games (observations up to the order of rows with equal coordinate 0),
    actions, rewards, dones and the state itself"""
    m, d = spec
    tag = f"fused_m{m}_d{d}_{role}"
    g = lambda k: live_fused[f"{tag}/{k}"]
    game = FusedGame(_SetNet(g("host_w"), g("host_b")), _SetNet(g("agent_w"), g("agent_b"), g("agent_v")),
                     log_time=True)
    pts = HipPoints(torch.as_tensor(g("start")))
    pts.get_newton_polytope()
    for t in range(3):
        obs, act, rew, done, nxt = game.step(pts, role, scale_observation=False, exploration_rate=0.0)
        o, n = (obs, nxt) if role == "host" else (obs["points"], nxt["points"])
        assert np.array_equal(_canonical(host(o)), _canonical(g(f"t{t}_obs"))), t
        assert np.array_equal(_canonical(host(n)), _canonical(g(f"t{t}_next"))), t
        assert bool((o[:, :-1, 0] >= o[:, 1:, 0]).all())
        if role == "agent":
            assert np.array_equal(host(obs["coords"]), g(f"t{t}_obs_coords")), t
            assert np.array_equal(host(nxt["coords"]), g(f"t{t}_next_coords")), t
        assert act.shape == (o.shape[0], 1) and np.array_equal(host(act), g(f"t{t}_actions")), t
        assert rew.dtype == torch.float32 and np.array_equal(host(rew), g(f"t{t}_rewards")), t
        assert done.dtype == torch.bool and np.array_equal(host(done), g(f"t{t}_dones")), t
        assert np.array_equal(host(pts.points), g(f"t{t}_state")), t
    assert "step-agent_move" in game.time_log
    with pytest.raises(TypeError):
        FusedGame(torch.nn.Identity(), torch.nn.Identity(), device="cpu")


def test_fused_game_types_and_exploration():
    """test/testTrainer.py:105-118: dtypes of the experiences for an f64 game; exploring players stay legal"""
    m, d, b = 20, 3, 256
    flat = torch.nn.Flatten()

    class AgentNet(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(m * d + d, d)

        def forward(self, x):
            return self.lin(torch.cat([flat(x["points"]), x["coords"]], dim=1))

    torch.manual_seed(0)
    game = FusedGame(torch.nn.Sequential(flat, torch.nn.Linear(m * d, 4)), AgentNet(), dtype=torch.float64,
                     log_time=False)
    pts = HipPoints(torch.randint(5, (b, m, d)), dtype=torch.float64)
    pts.get_newton_polytope()
    before = host(pts.get_num_points())
    exp = game.step(pts, "host", scale_observation=False)
    assert exp[0].dtype == torch.float64 and exp[1].dtype == torch.int32 and exp[2].dtype == torch.float32
    assert exp[3].dtype == torch.bool and exp[4].dtype == torch.float64
    assert exp[0].shape == (int((before >= 2).sum()), m, d)
    obs, act, rew, done, nxt = game.step(pts, "agent", scale_observation=True, exploration_rate=1.0)
    assert obs["coords"].shape[1] == d and float(nxt["points"].max()) <= 1.0
    chosen = torch.gather(obs["coords"], 1, act.long())
    assert act.shape[1] == 1 and chosen.shape == act.shape
    assert np.array_equal(host(rew), -host(done).astype(np.float32))


def test_policy_wrappers():
    """jax/util.py:153-169, 287-341; players.py:55-77,156-199; host_action_preprocess.py:38-52"""
    from hironaka_amd.functional import action_wrapper, apply_agent_action_mask, get_value_est_fn, mcts_wrapper
    from hironaka_amd.host_action_preprocess import decode, decode_from_one_hot
    from hironaka_amd.players import char_vector, choose_first_agent_fn_slice, choose_last_agent_fn_slice
    m, d = 4, 3
    obs = torch.zeros(5, m * d + d).cuda()
    obs[:, -d:] = dev([[1, 1, 0], [0, 1, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]])

    def agent_policy(x, scale=1.0):
        return scale * torch.arange(1, d + 1, device=x.device, dtype=torch.float32).expand(x.shape[0], d), torch.ones(x.shape[0])

    masked = apply_agent_action_mask(agent_policy, d, nan_free=True)
    pol, val = masked(obs, scale=2.0)
    assert host(pol).tolist()[0] == [2.0, 4.0, -np.inf] and host(pol).tolist()[1] == [-np.inf, 4.0, 6.0]
    literal, _ = apply_agent_action_mask(agent_policy, d)(obs)  # the reference's expression: inf * 0
    assert np.isnan(host(literal)[0, :2]).all() and host(literal)[0, 2] == -np.inf
    assert masked.__name__ == "agent_policy"
    act = action_wrapper(agent_policy, d)(obs)
    assert host(act).argmax(1).tolist() == [0, 1, 0, 0, 1]  # NaN wins the argmax: the first allowed axis
    assert host(action_wrapper(agent_policy)(obs)).argmax(1).tolist() == [2] * 5
    est = get_value_est_fn("agent")(None, torch.tensor([0, 1, 4]).cuda())
    assert host(est).tolist() == [-1.0, -1.0, -0.25]
    assert host(get_value_est_fn("host")(None, torch.tensor([2]).cuda())).tolist() == [0.5]
    assert host(choose_first_agent_fn_slice(obs[1], (m, d))).tolist() == [0, 1, 0]
    assert host(choose_last_agent_fn_slice(obs[0], (m, d))).tolist() == [0, 1, 0]
    assert host(char_vector(dev([5, 0, 0]), dev([0, 0, 3]))).tolist() == [8.0, 2.0]  # test/testZeillinger.py:19
    assert np.isinf(host(char_vector(dev([1, 1, 1]), dev([0, 0, 0])))).all()
    assert np.isinf(host(char_vector(dev([-1, -1, -1]), dev([0, 2, 0])))).all()
    table = decode_table(3)
    assert host(decode(2, table)).tolist() == [0, 1, 1]
    assert host(decode_from_one_hot(dev([0, 0, 1, 0]), table)).tolist() == [0, 1, 1]

    class _Out:
        action_weights = dev([[0.5, 0.5, 0.0]])

        class search_tree:
            node_values = dev([[0.25, 9.0]])

    pol, val = mcts_wrapper(lambda key, x, p, o: _Out)(None, None, None, 0)
    assert np.allclose(host(pol), np.log([[0.5, 0.5, 1e-8]])) and host(val).tolist() == [0.25]


def test_policy_players():
    """host.py:98-113, agent.py:101-111: players that ask a policy object, through GameHironaka"""
    from hironaka_amd.agent import PolicyAgent
    from hironaka_amd.host import PolicyHost

    class AllCoordPolicy:
        def predict(self, features):
            assert features.shape[1:] == (10, 3) and bool((features[:, :-1, 0] >= features[:, 1:, 0]).all())
            return torch.ones((features.shape[0], 3), device=features.device)

    class FirstAxisPolicy:
        def predict(self, inputs):
            features, coords = inputs
            return torch.argmax((coords > 0).to(torch.int32), dim=1).tolist()

    start = torch.randint(0, 20, (32, 10, 3), generator=torch.Generator().manual_seed(3))
    a = GameHironaka(HipPoints(start.clone()), PolicyHost(AllCoordPolicy()), PolicyAgent(FirstAxisPolicy()),
                     scale_observation=False)
    b = GameHironaka(HipPoints(start.clone()), AllCoordHost(), ChooseFirstAgent(), scale_observation=False)
    for _ in range(12):
        assert a.step() == b.step()
        assert torch.equal(a.state.points, b.state.points)
    assert len(a.move_history) == len(b.move_history) > 0
    assert torch.equal(a.coord_history[0], b.coord_history[0]) and torch.equal(a.move_history[0], b.move_history[0])


def test_rollout_values_kernel_random_patterns():
    """hk_rollout_values (the device path of rollout_postprocess) against the oracle's restatement of
    jax_trainer.py:558-592 / jax/util.py:261-284 on random observation patterns -- also ones no game produces (point
    counts that go up again: several finishing moves in one row), every role / tree kind, T = 1 .. 64"""
    rng = np.random.default_rng(11)
    d = 3
    for T in (1, 2, 5, 20, 33, 64):
        for role, unified, width in (("host", False, 60), ("agent", False, 63), ("host", True, 63), ("agent", True, 63)):
            b = 97
            obs = np.full((b, T, width), -1.0, dtype=np.float32)
            for i in range(b):
                monotone = rng.random() < 0.7
                n = int(rng.integers(1, 8))
                for t in range(T):
                    if monotone:
                        n = max(int(n - rng.integers(0, 2)), 0)
                    else:
                        n = int(rng.integers(0, 6))
                    k = n + (1 if width == 63 else 0)
                    obs[i, t, : k * d] = rng.integers(0, 5, k * d)
            rollouts = (dev(obs), torch.zeros(b, T, 4, device="cuda"), torch.zeros(b, T, device="cuda"))
            _, _, v = rollout_postprocess(rollouts, role, d, 0.99, unified)
            want = NO.rollout_postprocess(obs, d, 0.99, role, use_unified_tree=unified)
            # (rows with many finishing moves sum tens of terms: float32 accumulation order shows in the last bits)
            assert np.allclose(host(v), want, rtol=2e-6, atol=2e-6), (T, role, unified, np.abs(host(v) - want).max())
