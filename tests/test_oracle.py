"""Pin the oracle (numpy + C restatements) to the reference: hand-typed known answers from
the reference's own tests, outputs of the reference's torch/list siblings (live_*.npz), the
reference's native cppUtil (oracle/_ref, when built), and each other."""
import json
import os
import ctypes

import numpy as np
import pytest

from hironaka_amd import _abi as A
from oracle import c_oracle as CO
from oracle import np_oracle as NO

from conftest import ROOT


def f32(x):
    return np.array(x, dtype=np.float32)


class NumpyOracle:
    name = "numpy"
    shift = staticmethod(lambda p, c, a, pad=-1.0, sem="jax", noop=False, ign=False:
                         NO.shift(p, c, a, pad, sem, noop, ign))
    newton = staticmethod(lambda p, pad=-1.0, sem="jax", compact=False: NO.get_newton_polytope(p, pad, sem, compact))
    reposition = staticmethod(lambda p, pad=-1.0, sem="jax": NO.reposition(p, pad, sem))
    rescale = staticmethod(lambda p, pad=-1.0, sem="jax": NO.rescale(p, pad, sem))
    features = staticmethod(NO.get_features)
    zeillinger = staticmethod(NO.zeillinger_class)
    decode = staticmethod(lambda cls, d: NO.decode_class(cls, d))
    dones = staticmethod(NO.get_dones)
    num_points = staticmethod(NO.get_num_points)


class COracle:
    name = "c"
    shift = staticmethod(lambda p, c, a, pad=-1.0, sem="jax", noop=False, ign=False:
                         CO.shift(p, np.asarray(c).astype(p.dtype), np.asarray(a), pad, sem=sem,
                                  noop_if_invalid=noop, ignore_ended=ign))
    newton = staticmethod(lambda p, pad=-1.0, sem="jax", compact=False:
                          CO.get_newton_polytope(p, pad, sem=sem, compact_sorted=compact))
    reposition = staticmethod(lambda p, pad=-1.0, sem="jax": CO.reposition(p, pad, sem=sem))
    rescale = staticmethod(lambda p, pad=-1.0, sem="jax": CO.rescale(p, pad, sem=sem))
    features = staticmethod(CO.get_features)
    zeillinger = staticmethod(CO.zeillinger)
    decode = staticmethod(lambda cls, d: CO.decode_host_class(cls, d))
    dones = staticmethod(CO.get_dones)
    num_points = staticmethod(CO.get_num_points)


@pytest.fixture(params=[NumpyOracle, COracle], ids=["numpy", "c"])
def O(request):
    return request.param


# ------------------------------------------------------------------------------------------
# hand-typed vectors from the reference's tests
# ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("sem", ["jax", "torch"])
def test_reference_sequence_r_r2_r3_rs(O, golden, sem):
    """test/testJAX.py:80-101, test/testTensorPoints.py:48-82: newton -> shift+newton ->
    reposition -> rescale."""
    e = golden["newton_r"]
    r = O.newton(f32(e["points"]), sem=sem)
    assert np.array_equal(r, f32(e["expected"]))
    e2 = golden["shift_then_newton_r2"]
    r2 = O.newton(O.shift(r, np.array(e2["coords"]), np.array(e2["axis"]), sem=sem), sem=sem)
    assert np.array_equal(r2, f32(e2["expected"]))
    r3 = O.reposition(r2, sem=sem)
    assert np.array_equal(r3, f32(golden["reposition_r3"]["expected"]))
    rs = O.rescale(r3, sem=sem)
    assert np.allclose(rs, f32(golden["rescale_rs"]["expected"]), atol=1e-6)


def test_newton_edge_cases(O, golden):
    for key in ("newton_extreme", "torch_functions_2", "torch_remove_repeated"):
        e = golden[key]
        for sem in ("jax", "torch"):
            assert np.array_equal(O.newton(f32(e["points"]), sem=sem), f32(e["expected"])), key


def test_rescale_and_fractional_shift(O, golden):
    e = golden["rescale_s"]
    assert np.array_equal(O.rescale(f32(e["points"])), f32(e["expected"]))
    e = golden["shift_fractional"]
    assert np.array_equal(O.shift(f32(e["points"]), np.array(e["coords"]), np.array(e["axis"])), f32(e["expected"]))
    e = golden["torch_rescale_by_0"]
    assert np.isfinite(O.rescale(f32(e["points"]), sem="torch")).all()


def test_features(O, golden):
    e = golden["features_host"]
    assert np.array_equal(O.features(f32(e["points"]), True), f32(e["expected"]))
    for key in ("features_agent_unscaled", "features_agent_scaled"):
        e = golden[key]
        obs = f32(e["obs"])
        m, d = e["spec"]
        feat = O.features(obs[:, : m * d].reshape(-1, m, d).copy(), e["scale_observation"])
        assert np.array_equal(np.concatenate([feat, obs[:, m * d:]], axis=1), f32(e["expected"])), key


def test_codec(O, golden):
    e = golden["codec_d3"]
    assert np.array_equal(O.decode(np.arange(4, dtype=np.int32), 3), np.array(e["decode_table"]))
    assert np.array_equal(NO.decode_table(3), np.array(e["decode_table"]))
    assert np.array_equal(NO.encode_class(np.array(e["encode_in"])), np.array(e["encode_out"]))
    e = golden["host_action_encoder"]
    assert np.array_equal(NO.encode_class(np.array(e["masks"])), np.array(e["class"]))
    n = e["roundtrip_classes"]
    tab = O.decode(np.arange(n, dtype=np.int32), e["roundtrip_dim"])
    assert np.array_equal(NO.encode_class(tab), np.arange(n))
    e = golden["all_coord_host"]
    assert (O.decode(np.array(e["expected_class"], dtype=np.int32), 3) == 1).all()


def test_zeillinger_known_answers(O, golden):
    e = golden["zeillinger_slice"]
    assert O.zeillinger(f32(e["points"])).tolist() == e["expected_class"]
    assert NO.zeillinger_list(f32(e["points"])[0]) == tuple(e["expected_list_coords"][0])
    e = golden["zeillinger_batch"]
    assert np.array_equal(O.decode(O.zeillinger(f32(e["points"])), 4), np.array(e["expected_mask"]))
    e = golden["zeillinger_padded"]
    assert O.zeillinger(f32(e["points"])).tolist() == e["expected_class"]
    e = golden["zeillinger_ignore_batch"]
    assert NO.zeillinger_list(f32(e["points"])[0]) == (0, 2)
    v = np.array(golden["zeillinger_char_vector"]["vector"])
    assert [v.max() - v.min(), int((v == v.max()).sum() + (v == v.min()).sum())] == golden["zeillinger_char_vector"]["expected"]


def test_torch_mode_noops(O, golden):
    """test/testTensorPoints.py:84-103: illegal axis and finished games are not shifted."""
    e = golden["torch_invalid_actions"]
    mask = np.array([[0, 1, 0, 0], [1, 0, 1, 1]])
    out = O.shift(f32(e["points"]), mask, np.array(e["axis"]), sem="torch", noop=True, ign=True)
    assert np.array_equal(out, f32(e["expected"]))
    e = golden["torch_ended_game"]
    mask = np.array([[1, 1, 0, 0]])
    assert np.array_equal(O.shift(f32(e["points"]), mask, np.array(e["axis"]), sem="torch", noop=True, ign=True),
                          f32(e["expected_ignore_ended"]))
    assert np.array_equal(O.shift(f32(e["points"]), mask, np.array(e["axis"]), sem="torch", noop=True, ign=False),
                          f32(e["expected_forced"]))


def test_list_mode_known_answers(O, golden):
    e = golden["list_shift"]
    mask = np.array([[0, 0, 1, 1]])
    assert np.array_equal(O.shift(f32(e["points"]), mask, np.array(e["axis"]), sem="list", noop=True), f32(e["expected"]))
    e = golden["list_newton"]
    r = O.newton(f32(e["points"]), sem="list")
    assert np.array_equal(r[0, :5], f32(e["expected_compact"])[0]) and (r[0, 5:] == -1).all()
    e = golden["list_reposition"]
    assert np.array_equal(O.reposition(f32(e["points"]), sem="list"), f32(e["expected"]))
    e = golden["list_rescale"]
    assert np.allclose(O.rescale(np.array(e["points"], dtype=np.float64), sem="list"), np.array(e["expected"]), rtol=0, atol=1e-15)
    e = golden["list_choose_first_move"]
    p = O.shift(f32(e["points"]), np.array([[1, 1]]), np.array(e["axis"]), sem="list", noop=True)
    r = O.newton(p, sem="list")
    assert np.array_equal(r[0, :1], f32(e["expected_compact"])[0]) and (r[0, 1:] == -1).all()


def test_reward_and_take_actions_composition(golden):
    e = golden["reward_convention"]
    done, prev = np.array(e["done"]), np.array(e["prev_done"])
    assert np.array_equal(NO.reward(done, prev, "host"), f32(e["host"]))
    assert np.array_equal(NO.reward(done, prev, "agent"), f32(e["agent"]))
    with pytest.raises(ValueError):
        NO.reward(done, prev, "referee")
    # test/testJAX.py:461-488
    e = golden["take_actions_inputs"]
    p, c, a = f32(e["points"]), np.array(e["coords"]), f32(e["axis"])
    host = CO.step(p, c.astype(np.float32), a, stages=A.HK_STAGE_SHIFT | A.HK_STAGE_NEWTON | A.HK_STAGE_RESCALE)
    assert np.array_equal(host["points"], NO.rescale(NO.get_newton_polytope(NO.shift(p, c, a))))
    obs = NO.make_agent_obs(p, c)
    agent = CO.step(obs, None, a, stages=A.HK_STAGE_SHIFT | A.HK_STAGE_NEWTON, coords_kind=A.HK_COORDS_IN_RECORD,
                    max_points=4, dim=3, reward_sign=-1.0)
    assert np.array_equal(agent["points"], NO.get_newton_polytope(NO.shift(p, c, a)))
    assert (agent["reward"] <= 0).all()


# ------------------------------------------------------------------------------------------
# outputs of the reference's torch sibling
# ------------------------------------------------------------------------------------------

def _torch_tags(live):
    return sorted({k.split("/")[0] for k in live.files if k.startswith("m")})


def test_live_torch_operators(O, live_torch):
    tags = _torch_tags(live_torch)
    assert len(tags) >= 20
    for tag in tags:
        g = lambda k: live_torch[f"{tag}/{k}"]
        p = g("points")
        pad = -1.0 if tag.endswith("pad1") else -1e-8
        assert np.array_equal(O.newton(p, pad, "torch"), g("newton")), tag
        assert np.array_equal(O.reposition(p, pad, "torch"), g("reposition")), tag
        assert np.array_equal(O.rescale(p, pad, "torch"), g("rescale")), tag
        mask, axis = g("mask"), g("axis")
        assert np.array_equal(O.decode(g("class"), p.shape[2]), mask.astype(np.int32)), tag
        for ign in (0, 1):
            out = O.shift(p, mask, axis, pad, "torch", noop=True, ign=bool(ign))
            assert np.array_equal(out, g(f"shift_ign{ign}")), (tag, ign)
        # FusedGame.agent_move sequence
        start = O.newton(p, pad, "torch")
        assert np.array_equal(start, g("game_start")), tag
        un = O.newton(O.shift(start, mask, axis, pad, "torch", noop=True, ign=True), pad, "torch")
        assert np.array_equal(un, g("game_unscaled")), tag
        assert np.array_equal(O.dones(un), g("game_ended")), tag
        assert np.array_equal(O.num_points(un), g("game_num_points")), tag
        assert np.array_equal(O.rescale(un, pad, "torch"), g("game_scaled")), tag


def test_live_torch_features(live_fused):
    """TensorPoints.get_features as run by the reference (distinct first coordinates: the inputs on which its
    unstable argsort is defined) == the restatement; plus the restatement's own tie rule (row order)"""
    from oracle import c_oracle as CO
    tags = sorted({k.split("/")[0] for k in live_fused.files if k.startswith("features_")})
    assert len(tags) == 4
    for tag in tags:
        p = live_fused[f"{tag}/points"]
        assert np.array_equal(CO.get_features_torch(p), live_fused[f"{tag}/features"]), tag
        assert np.array_equal(CO.get_features_torch(p.astype(np.float64)), live_fused[f"{tag}/features"]), tag
    p = np.array([[[1, 7], [2, 0], [-1, -1], [1, 3], [2, 5], [0, 9]]], dtype=np.float32)
    want = np.array([[[2, 0], [2, 5], [1, 7], [1, 3], [0, 9], [-1, -1]]], dtype=np.float32)
    assert np.array_equal(CO.get_features_torch(p), want)


def test_live_torch_jax_agreement(O, live_torch):
    """On legal moves of unfinished games with padding -1 the JAX semantics must give the
    torch sibling's values."""
    for tag in _torch_tags(live_torch):
        if not tag.endswith("pad1"):
            continue
        g = lambda k: live_torch[f"{tag}/{k}"]
        start, mask, axis = g("game_start"), g("mask"), g("axis")
        legal = mask[np.arange(len(axis)), axis] == 1
        alive = ~O.dones(start)
        sel = legal & alive
        assert sel.sum() > 0
        out = O.newton(O.shift(start, mask, axis, -1.0, "jax"), -1.0, "jax")
        assert np.array_equal(out[sel], g("game_unscaled")[sel]), tag
        assert np.array_equal(O.newton(g("points"), -1.0, "jax"), g("newton")), tag


def test_live_codec_tables(O, live_torch):
    for d in range(2, 8):
        n = 2 ** d - d - 1
        assert np.array_equal(O.decode(np.arange(n, dtype=np.int32), d), live_torch[f"codec/d{d}"])
        assert np.array_equal(NO.encode_class(live_torch[f"codec/d{d}"]), live_torch[f"codec/d{d}_roundtrip"])


# ------------------------------------------------------------------------------------------
# outputs of the reference's list sibling, incl. BASELINE config 1 trajectories
# ------------------------------------------------------------------------------------------

def test_live_list_operators(O, live_list):
    for tag in ("m6_d4", "m10_d3", "m20_d3", "m5_d2"):
        g = lambda k: live_list[f"{tag}/{k}"]
        p = g("points")
        assert np.array_equal(O.newton(p, sem="list"), g("newton")), tag
        assert np.array_equal(O.reposition(p, sem="list"), g("reposition")), tag
        assert np.array_equal(O.rescale(p, sem="list"), g("rescale")), tag
        sh = O.shift(p, g("shift_mask"), g("shift_axis"), sem="list", noop=True)
        assert np.array_equal(sh, g("shift")), tag
        assert np.array_equal(O.newton(sh, sem="list"), g("shift_newton")), tag
        # same survivor SET under the JAX semantics (in-place order)
        jx = O.newton(p, sem="jax")
        for b in range(len(p)):
            a = {tuple(r) for r in jx[b] if r[0] >= 0}
            c = {tuple(r) for r in g("newton")[b] if r[0] >= 0}
            assert a == c


@pytest.mark.parametrize("scale", [0, 1])
def test_live_config1_trajectory(O, live_list, scale):
    """BASELINE config 1: dim 3, 10 points, 32 games, Zeillinger host vs (recorded) random
    agent through GameHironaka.step (game.py:87-119): replay and compare every state."""
    states = live_list[f"game_scale{scale}/states"]
    masks, axes = live_list[f"game_scale{scale}/masks"], live_list[f"game_scale{scale}/axes"]
    start = live_list[f"game_scale{scale}/start"]
    p = O.newton(start, sem="list")
    if scale:
        p = O.rescale(p, sem="list")
    assert np.array_equal(p, states[0])
    assert len(masks) >= 3
    for t in range(len(masks)):
        # the host's choice is Zeillinger's (host.py:70-95) on the compacted state
        for b in range(len(p)):
            rows = p[b][p[b][:, 0] >= 0]
            if len(rows) >= 2:
                lo, hi = NO.zeillinger_list(rows)
                want = np.zeros(3, dtype=np.int32)
                want[[lo, hi]] = 1
                assert np.array_equal(want, masks[t][b]), (t, b)
        p = O.newton(O.shift(p, masks[t], axes[t], sem="list", noop=True), sem="list")
        if scale:
            p = O.rescale(p, sem="list")
        assert np.array_equal(p, states[t + 1]), t
    assert O.dones(p).all()


def test_live_zeillinger_list(live_list):
    pts, coords = live_list["zeillinger/points"], live_list["zeillinger/coords"]
    for b in range(len(pts)):
        rows = pts[b][pts[b][:, 0] >= 0]
        assert NO.zeillinger_list(rows) == tuple(coords[b]), b
    # the C restatement of the list-semantics host (class id of the same subset; -1 below 2 points)
    cls = CO.zeillinger(pts, sem="list")
    want = np.zeros((len(pts), pts.shape[2]), dtype=np.int32)
    for b in range(len(pts)):
        want[b, coords[b]] = 1
    assert np.array_equal(NO.decode_class(cls, pts.shape[2]), want)
    lone = np.full((2, 4, 3), -1.0)
    lone[1, 0] = [1, 2, 3]
    assert CO.zeillinger(lone, sem="list").tolist() == [-1, -1]
    for scale in (0, 1):  # and the host's choices inside the recorded config-1 games
        states, masks = live_list[f"game_scale{scale}/states"], live_list[f"game_scale{scale}/masks"]
        for t in range(len(masks)):
            cls = CO.zeillinger(states[t], sem="list")
            alive = cls >= 0
            assert np.array_equal(NO.decode_class(cls[alive], 3), masks[t][alive]), (scale, t)


# ------------------------------------------------------------------------------------------
# the reference's native routine (oracle/_ref/cppUtil.so), where it is correct
# ------------------------------------------------------------------------------------------

def test_native_reference_cpputil():
    """hironaka/cpp/cppUtil.cpp:58-61 getNewtonPolytope_approx compiled as-is.  It is only
    right for games with >= 2 points and no duplicates; there its compacted,
    original-order survivor list must equal the oracle's survivors."""
    path = os.path.join(ROOT, "oracle", "_ref", "cppUtil.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    ref = ctypes.CDLL(path)
    ref.getNewtonPolytope_approx.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    ref.getNewtonPolytope_approx.restype = None
    rng = np.random.default_rng(5)
    b, m, d = 200, 8, 3
    pts = np.full((b, m, d), -1, dtype=np.int64)
    for g in range(b):
        n = int(rng.integers(2, m + 1))
        rows = set()
        while len(rows) < n:
            rows.add(tuple(rng.integers(0, 9, d).tolist()))
        pts[g, :n] = np.array(sorted(rows, key=lambda _: rng.random()))
    out = np.full_like(pts, -1)
    ref.getNewtonPolytope_approx(pts.ctypes.data, b, m, d, out.ctypes.data)
    mine = CO.get_newton_polytope(pts.astype(np.float64), sem="jax")
    for g in range(b):
        want = [tuple(r) for r in out[g] if r[0] != -1]
        got = [tuple(int(v) for v in r) for r in mine[g] if r[0] >= 0]
        assert want == got, g


# ------------------------------------------------------------------------------------------
# the two restatements against each other (all modes, both dtypes), rollouts, generator
# ------------------------------------------------------------------------------------------

def _rand_state(rng, b, m, d, dtype, pad):
    p = rng.integers(0, 6, (b, m, d)).astype(dtype)
    p[rng.random((b, m)) < 0.3] = pad
    return p


@pytest.mark.parametrize("spec", [(4, 3), (5, 2), (10, 3), (20, 3), (7, 4), (6, 5)])
def test_numpy_vs_c_all_modes(spec):
    rng = np.random.default_rng(hash(spec) % 1000)
    m, d = spec
    for dtype in (np.float32, np.float64):
        for sem in ("jax", "torch", "list"):
            for pad in (-1.0, -1e-8):
                p = _rand_state(rng, 48, m, d, dtype, pad)
                cls = rng.integers(0, 2 ** d - d - 1, 48).astype(np.int32)
                mask, ax = NO.decode_class(cls, d), rng.integers(0, d, 48).astype(np.int32)
                for compact in (False, True):
                    assert np.array_equal(NO.get_newton_polytope(p, pad, sem, compact),
                                          CO.get_newton_polytope(p, pad, sem=sem, compact_sorted=compact))
                assert np.array_equal(NO.reposition(p, pad, sem), CO.reposition(p, pad, sem=sem))
                with np.errstate(all="ignore"):
                    assert np.array_equal(NO.rescale(p, pad, sem), CO.rescale(p, pad, sem=sem))
                for noop in (False, True):
                    for ign in (False, True):
                        assert np.array_equal(NO.shift(p, mask, ax, pad, sem, noop, ign),
                                              CO.shift(p, cls, ax.astype(np.float32), pad, sem=sem,
                                                       noop_if_invalid=noop, ignore_ended=ign))
                with np.errstate(all="ignore"):
                    want = NO.step(p, mask, ax, padding_value=pad, sem=sem, do_reposition=True, do_rescale=True)
                got = CO.step(p, cls, ax, stages=15, flags=CO.flags_of(sem=sem), padding_value=pad)
                assert np.array_equal(want, got["points"])
                assert np.array_equal(got["done"], NO.get_dones(want))
                assert np.array_equal(got["prev_done"], NO.get_dones(p))
                assert np.array_equal(got["reward"], NO.reward(got["done"], got["prev_done"]))
                assert np.array_equal(got["num_points"], NO.get_num_points(want))
    pf = _rand_state(rng, 48, m, d, np.float32, -1.0)
    assert np.array_equal(NO.zeillinger_class(pf), CO.zeillinger(pf))
    with np.errstate(all="ignore"):
        assert np.array_equal(NO.get_features(pf, True), CO.get_features(pf, True))
    assert np.array_equal(NO.get_features(pf, False), CO.get_features(pf, False))


@pytest.mark.parametrize("spec", [(10, 3), (20, 3), (8, 4)])
def test_generate_and_rollout_numpy_vs_c(spec):
    m, d = spec
    g1 = NO.generate_points(40, m, d, 20, seed=42, game_offset=5)
    assert np.array_equal(g1, CO.generate_points(40, m, d, 20, 42, 5))
    assert g1.max() < 20 and (NO.get_num_points(g1) >= 1).all()
    for hp in (0, 1, 2):
        for ap in (0, 1, 2, 3):
            pa, ra = NO.rollout(g1, 6, 123, game_offset=3, step_offset=2, host_policy=hp, agent_policy=ap,
                                padding_value=-1.0, sem="jax", do_reposition=True)
            pc, rc = CO.rollout(g1, 6, 123, game_offset=3, step_offset=2, host_policy=hp, agent_policy=ap)
            assert np.array_equal(pa, pc)
            for k in ("obs", "host_class", "axis", "done", "reward", "done_count", "game_length"):
                assert np.array_equal(ra[k], rc[k]), k


def test_rollout_sharding_invariance():
    """A shard with game_offset equals the matching slice of the unsharded run (the multi-GPU
    contract of SURVEY.md 8e)."""
    full = CO.generate_points(64, 20, 3, 20, 9)
    parts = [CO.generate_points(16, 20, 3, 20, 9, game_offset=16 * r) for r in range(4)]
    assert np.array_equal(full, np.concatenate(parts))
    pf, rf = CO.rollout(full, 8, 77)
    for r in range(4):
        pp, rp = CO.rollout(parts[r], 8, 77, game_offset=16 * r)
        assert np.array_equal(pp, pf[16 * r:16 * r + 16])
        assert np.array_equal(rp["axis"], rf["axis"][:, 16 * r:16 * r + 16])
    assert rf["done_count"][-1] == rf["done"][-1].sum()


def test_rollout_game_ids_permutation_invariance():
    """hk_rollout_desc.game_ids (ABI 3): a re-ordered batch with the permutation as game ids is, game by game, the
    run of the original order -- records, lengths and counts (the contract `ops.bin_by_live_rows` relies on)."""
    p0 = CO.generate_points(97, 20, 3, 20, 4)
    perm = np.argsort(-(p0[:, :, 0] >= 0).sum(1), kind="stable").astype(np.int32)
    for hp, ap in ((A.HK_HOST_RANDOM, A.HK_AGENT_RANDOM), (A.HK_HOST_ZEILLINGER, A.HK_AGENT_RANDOM_LEGAL)):
        pf, rf = CO.rollout(p0, 26, 3, game_offset=1 << 33, step_offset=2, host_policy=hp, agent_policy=ap)
        pb, rb = CO.rollout(p0[perm], 26, 3, game_offset=1 << 33, step_offset=2, host_policy=hp, agent_policy=ap,
                            game_ids=perm)
        assert np.array_equal(pb, pf[perm]) and np.array_equal(rb["game_length"], rf["game_length"][perm])
        for k in ("obs", "host_class", "axis", "done", "reward"):
            assert np.array_equal(rb[k], rf[k][:, perm]), k
        assert np.array_equal(rb["done_count"], rf["done_count"])


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors):
    counter = key = 0 and the pi-digits vector."""
    out = NO.philox4x32(0, 0, 0, 0, 0)
    assert [int(x) for x in out] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    out = NO.philox4x32(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, (0x299F31D0 << 32) | 0xA4093822)
    assert [int(x) for x in out] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def _postprocess_cases():
    with open(os.path.join(ROOT, "tests", "golden", "rollout_postprocess.json")) as f:
        doc = json.load(f)
    return doc["dimension"], doc["discount"], doc["cases"]


def test_rollout_postprocess_reference_vectors():
    """the value targets of JAXTrainer.rollout_postprocess, against the literal arrays of
    the reference's test_rollout_postprocess (test/testJAXTrainer.py:91-389, extracted as data by
    tests/golden/extract_postprocess_vectors.py).  Oracle and the product's torch plumbing both."""
    import torch
    from hironaka_amd.rollout import rollout_postprocess

    dimension, discount, cases = _postprocess_cases()
    assert len(cases) == 5
    for case in cases:
        obs = np.asarray(case["obs"], dtype=np.float32)
        want = np.asarray(case["expected"], dtype=np.float32).ravel()
        got = NO.rollout_postprocess(obs, dimension, discount, case["role"], case["unified"])
        assert got.shape == want.shape and np.allclose(got, want, rtol=0, atol=1e-6)
        b, t, w = obs.shape
        tobs = torch.from_numpy(obs)
        o, p, v = rollout_postprocess((tobs, torch.zeros(b, t, 4), torch.zeros(b, t)), case["role"], dimension,
                                      discount, case["unified"])
        assert o.shape == (b * t, w) and p.shape == (b * t, 4) and v.shape == (b * t,)
        assert np.allclose(v.numpy(), want, rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        NO.rollout_postprocess(obs, dimension, discount, "referee", False)


def test_torch_cpu_array_formulation_equals_numpy():
    """oracle/torch_cpu_oracle.py (bench.py's all-core "reference CPU algorithm" leg) against np_oracle, bit for
    bit: single steps on random holes / duplicates and a 12-step random-policy rollout."""
    import torch
    from oracle import torch_cpu_oracle as TO
    rng = np.random.default_rng(5)
    for m, d in ((20, 3), (10, 3), (12, 4)):
        p = rng.integers(0, 6, (300, m, d)).astype(np.float32)
        p[rng.random((300, m)) < 0.4] = -1.0
        cls = rng.integers(0, 2 ** d - d - 1, 300)
        ax = rng.integers(0, d, 300)
        want = NO.step(p, NO.decode_class(cls, d), ax)
        got = TO.step(torch.from_numpy(p), torch.from_numpy(NO.decode_class(cls, d)).float(),
                      torch.from_numpy(ax)).numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    fresh = NO.generate_points(256, 20, 3, 20, 42)
    want_p, want_rec = NO.rollout(fresh, 12, 7)
    got_p, got_counts = TO.rollout(fresh, 12, 7)
    assert np.array_equal(got_p, want_p) and np.array_equal(got_counts, want_rec["done_count"])
