"""
Generate tests/golden/live_*.npz by RUNNING the reference's own torch / list siblings of the
hot path on seeded random inputs.  Runs only where /root/reference exists (the build
container); the resulting .npz files are plain input/output arrays and are what travels.

The JAX path itself cannot be imported here (jax/chex/flax/mctx/gym are not installed:
ordinary ModuleNotFoundError), so the jax-free files are loaded one by one
under synthetic `hironaka`, `hironaka.src`, `hironaka.core` package objects, bypassing the
package __init__ files that import jax:

    hironaka/src/_fn.py, _list_ops.py, _torch_ops.py
    hironaka/core/points_base.py, list_points.py, tensor_points.py
    hironaka/points.py, host.py, agent.py, game.py, policy/policy.py

    hironaka/trainer/timer.py, fused_game.py   (live_fused_game.npz)

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [torch list fused]
"""
import importlib.util
import os
import random
import sys
import types

sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference tree

import numpy as np
import torch

REF = os.environ.get("HIRONAKA_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    hir = _pkg("hironaka")
    src = _pkg("hironaka.src")
    core = _pkg("hironaka.core")
    pol = _pkg("hironaka.policy")
    fn = _load("hironaka.src._fn", "hironaka/src/_fn.py")
    for k in dir(fn):
        if not k.startswith("__"):
            setattr(src, k, getattr(fn, k))
    lst = _load("hironaka.src._list_ops", "hironaka/src/_list_ops.py")
    for k in ("get_newton_polytope_approx_lst", "get_newton_polytope_lst", "reposition_lst", "shift_lst"):
        setattr(src, k, getattr(lst, k))
    tor = _load("hironaka.src._torch_ops", "hironaka/src/_torch_ops.py")
    for k in ("get_newton_polytope_torch", "get_newton_polytope_approx_torch", "reposition_torch",
              "rescale_torch", "shift_torch"):
        setattr(src, k, getattr(tor, k))
    pb = _load("hironaka.core.points_base", "hironaka/core/points_base.py")
    lp = _load("hironaka.core.list_points", "hironaka/core/list_points.py")
    tp = _load("hironaka.core.tensor_points", "hironaka/core/tensor_points.py")
    core.PointsBase, core.ListPoints, core.TensorPoints = pb.PointsBase, lp.ListPoints, tp.TensorPoints
    pp = _load("hironaka.policy.policy", "hironaka/policy/policy.py")
    pol.Policy = pp.Policy
    pts = _load("hironaka.points", "hironaka/points.py")
    host = _load("hironaka.host", "hironaka/host.py")
    agent = _load("hironaka.agent", "hironaka/agent.py")
    game = _load("hironaka.game", "hironaka/game.py")
    return types.SimpleNamespace(fn=fn, lst=lst, tor=tor, ListPoints=lp.ListPoints,
                                 TensorPoints=tp.TensorPoints, host=host, agent=agent, game=game)


def random_states(rng, b, m, d, max_value, dtype, pad=-1.0, holes=0.25):
    p = rng.integers(0, max_value, (b, m, d)).astype(dtype)
    p[rng.random((b, m)) < holes] = pad
    return p


def pad_lists(games, m, d, pad=-1.0):
    out = np.full((len(games), m, d), pad, dtype=np.float64)
    for g, rows in enumerate(games):
        if len(rows):
            out[g, : len(rows)] = np.array(rows, dtype=np.float64)
    return out


def make_torch(ref, rng):
    """Per spec: inputs and the outputs of every torch operator, incl. illegal moves, finished
    games, duplicate rows, custom padding values and the FusedGame.agent_move sequence."""
    rec = {}
    specs = [(4, 3), (5, 2), (10, 3), (20, 3), (8, 4), (50, 4), (6, 5)]
    for (m, d) in specs:
        for dt_name, tdt, ndt in (("f32", torch.float32, np.float32), ("f64", torch.float64, np.float64)):
            for pad in (-1.0, -1e-8):
                if dt_name == "f64" and m > 10:
                    continue
                tag = f"m{m}_d{d}_{dt_name}_pad{'1' if pad == -1.0 else 'eps'}"
                b = 48 if m <= 20 else 12
                p = random_states(rng, b, m, d, 7, ndt, pad)
                p[0, :] = pad  # an empty game
                p[1, 1:] = pad  # a one-point game
                p[1, 0] = np.abs(p[1, 0]) + 1
                t = torch.tensor(p)
                rec[tag + "/points"] = p
                rec[tag + "/newton"] = ref.tor.get_newton_polytope_torch(t.clone(), inplace=False, padding_value=pad).numpy()
                rec[tag + "/reposition"] = ref.tor.reposition_torch(t.clone(), inplace=False, padding_value=pad).numpy()
                rec[tag + "/rescale"] = ref.tor.rescale_torch(t.clone(), inplace=False, padding_value=pad).numpy()
                n_cls = 2 ** d - d - 1
                enc = ref.fn.HostActionEncoder(d)
                cls = rng.integers(0, n_cls, b).astype(np.int64)
                mask = enc.decode_tensor(torch.tensor(cls), dtype=tdt)
                axis = rng.integers(0, d, b).astype(np.int64)  # legal and illegal moves mixed
                rec[tag + "/class"] = cls.astype(np.int32)
                rec[tag + "/mask"] = mask.numpy()
                rec[tag + "/axis"] = axis.astype(np.int32)
                for ign in (True, False):
                    rec[tag + f"/shift_ign{int(ign)}"] = ref.tor.shift_torch(
                        t.clone(), mask, torch.tensor(axis), inplace=False, padding_value=pad,
                        ignore_ended_games=ign).numpy()
                # FusedGame.agent_move (trainer/fused_game.py:150-163): shift -> newton -> rescale
                tp = ref.TensorPoints(t.clone(), padding_value=pad, dtype=tdt)
                tp.get_newton_polytope()
                rec[tag + "/game_start"] = tp.points.numpy().copy()
                tp.shift(mask, torch.tensor(axis).type(tdt))
                tp.get_newton_polytope()
                rec[tag + "/game_unscaled"] = tp.points.numpy().copy()
                rec[tag + "/game_ended"] = tp.ended_batch_in_tensor.numpy().copy()
                rec[tag + "/game_num_points"] = tp.get_num_points().numpy().astype(np.int32)
                tp.rescale()
                rec[tag + "/game_scaled"] = tp.points.numpy().copy()
    # codec tables (src/_fn.py:241-325)
    for d in range(2, 8):
        enc = ref.fn.HostActionEncoder(d)
        n_cls = 2 ** d - d - 1
        rec[f"codec/d{d}"] = enc.decode_tensor(torch.arange(n_cls)).numpy().astype(np.int32)
        rec[f"codec/d{d}_roundtrip"] = enc.encode_tensor(enc.decode_tensor(torch.arange(n_cls))).numpy().astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "live_torch.npz"), **rec)
    return len(rec)


def make_list(ref, rng):
    """ListPoints operators on ragged integer games (padded at the boundary), Zeillinger's
    choices, and full GameHironaka trajectories (BASELINE config 1: dim 3, 10 points, 32 games,
    Zeillinger vs RandomAgent; the agent's choices are recorded so the replay is data)."""
    rec = {}
    for (m, d) in [(6, 4), (10, 3), (20, 3), (5, 2)]:
        tag = f"m{m}_d{d}"
        b = 32
        games = [rng.integers(0, 9, (int(rng.integers(1, m + 1)), d)).tolist() for _ in range(b)]
        rec[tag + "/points"] = pad_lists(games, m, d)
        lp = ref.ListPoints([[list(r) for r in g] for g in games])
        lp.get_newton_polytope()
        rec[tag + "/newton"] = pad_lists(lp.points, m, d)
        lp2 = ref.ListPoints([[list(r) for r in g] for g in games])
        lp2.reposition()
        rec[tag + "/reposition"] = pad_lists(lp2.points, m, d)
        lp3 = ref.ListPoints([[list(r) for r in g] for g in games])
        lp3.rescale()
        rec[tag + "/rescale"] = pad_lists(lp3.points, m, d)
        coords, axis = [], []
        for _ in range(b):
            k = int(rng.integers(2, d + 1))
            c = sorted(rng.choice(d, size=k, replace=False).tolist())
            coords.append(c)
            axis.append(int(rng.integers(0, d)))  # may be illegal -> no-op
        lp4 = ref.ListPoints([[list(r) for r in g] for g in games])
        lp4.shift(coords, axis)
        rec[tag + "/shift"] = pad_lists(lp4.points, m, d)
        rec[tag + "/shift_mask"] = ref.fn.batched_coord_list_to_binary(coords, d).astype(np.int32)
        rec[tag + "/shift_axis"] = np.array(axis, dtype=np.int32)
        lp4.get_newton_polytope()
        rec[tag + "/shift_newton"] = pad_lists(lp4.points, m, d)

    # ---- config 1 trajectories -----------------------------------------------------------
    random.seed(0)
    np.random.seed(0)
    for scale in (False, True):
        m, d, b = 10, 3, 32
        tag = f"game_scale{int(scale)}"
        start = np.random.randint(0, 20, (b, m, d)).tolist()
        rec[tag + "/start"] = np.array(start, dtype=np.float64)
        state = ref.ListPoints(start, value_threshold=1e8)
        game = ref.game.GameHironaka(state, ref.host.Zeillinger(), ref.agent.RandomAgent(),
                                     scale_observation=scale)
        states = [pad_lists(game.state.points, m, d)]
        masks, axes = [], []
        alive = not game.stopped
        while alive:
            alive = game.step()
            coords, action = game.coord_history[-1], game.move_history[-1]
            mk = np.zeros((b, d), dtype=np.int32)
            ax = np.full(b, -1, dtype=np.int32)
            for g in range(b):
                mk[g, coords[g]] = 1
                if action[g] is not None:
                    ax[g] = action[g]
            masks.append(mk)
            axes.append(ax)
            states.append(pad_lists(game.state.points, m, d))
        rec[tag + "/states"] = np.stack(states)
        rec[tag + "/masks"] = np.stack(masks)
        rec[tag + "/axes"] = np.stack(axes)

    # ---- Zeillinger on compacted states (host.py:54-95) -------------------------------------
    zp, zc = [], []
    host = ref.host.Zeillinger()
    m, d = 8, 4
    for _ in range(64):
        rows = rng.integers(0, 12, (int(rng.integers(2, m + 1)), d)).tolist()
        lp = ref.ListPoints([rows])
        lp.get_newton_polytope()
        if len(lp.points[0]) < 2:
            continue
        zp.append(pad_lists(lp.points, m, d)[0])
        c = host.select_coord(lp)[0]
        zc.append([int(c[0]), int(c[1])])
    rec["zeillinger/points"] = np.stack(zp)
    rec["zeillinger/coords"] = np.array(zc, dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "live_list.npz"), **rec)
    return len(rec)


class SetNet(torch.nn.Module):
    """A row-permutation-invariant linear 'network' with exactly representable arithmetic (small integer
    weights, distinct dyadic biases): logits = sum_i rows_i @ W (+ coords @ V) + bias.  The reference orders
    rows with equal coordinate 0 however torch.argsort happens to; an invariant net makes the game's
    trajectory independent of that.  tests/test_gpu_surfaces.py rebuilds the same module from the arrays."""

    def __init__(self, w, bias, v=None):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor(w, dtype=torch.float32), requires_grad=False)
        self.bias = torch.nn.Parameter(torch.tensor(bias, dtype=torch.float32), requires_grad=False)
        self.v = None if v is None else torch.nn.Parameter(torch.tensor(v, dtype=torch.float32), requires_grad=False)

    def forward(self, x):
        if isinstance(x, dict):
            return x["points"].sum(dim=1) @ self.w + x["coords"] @ self.v + self.bias
        return x.sum(dim=1) @ self.w + self.bias


def make_fused_game(ref, rng):
    """TensorPoints.get_features (tensor_points.py:72-74) on games whose live rows have distinct first
    coordinates (the only inputs on which the reference's unstable argsort is defined), and
    FusedGame.step (trainer/fused_game.py:54-102) for both roles with the SetNet players above,
    exploration off, three consecutive steps per role."""
    _pkg("hironaka.trainer")
    _load("hironaka.trainer.timer", "hironaka/trainer/timer.py")
    sys.modules["hironaka.src"].HostActionEncoder = ref.fn.HostActionEncoder
    fg = _load("hironaka.trainer.fused_game", "hironaka/trainer/fused_game.py")
    rec = {}
    for (m, d) in [(5, 3), (20, 3), (8, 4), (12, 5)]:
        tag = f"features_m{m}_d{d}"
        b = 40
        p = rng.integers(0, 50, (b, m, d)).astype(np.float32)
        for g in range(b):
            p[g, :, 0] = rng.permutation(60)[:m]  # distinct first coordinates
        p[rng.random((b, m)) < 0.4] = -1.0
        p[0] = -1.0
        rec[tag + "/points"] = p
        rec[tag + "/features"] = ref.TensorPoints(torch.tensor(p), padding_value=-1.0).get_features().numpy()
    for (m, d) in [(20, 3), (8, 4)]:
        n_cls = 2 ** d - d - 1
        for role in ("host", "agent"):
            for scale in (False,):
                tag = f"fused_m{m}_d{d}_{role}"
                b = 64
                hw = rng.integers(-1, 2, (d, n_cls)).astype(np.float32)
                hb = (np.arange(n_cls) / 64.0).astype(np.float32)
                aw = rng.integers(-1, 2, (d, d)).astype(np.float32)
                av = rng.integers(-1, 2, (d, d)).astype(np.float32)
                ab = (np.arange(d)[::-1] / 64.0).astype(np.float32)
                start = rng.integers(0, 5, (b, m, d)).astype(np.float32)
                start[rng.random((b, m)) < 0.3] = -1.0
                start[0] = -1.0
                start[1, 1:] = -1.0
                rec[tag + "/host_w"], rec[tag + "/host_b"] = hw, hb
                rec[tag + "/agent_w"], rec[tag + "/agent_v"], rec[tag + "/agent_b"] = aw, av, ab
                rec[tag + "/start"] = start
                game = fg.FusedGame(SetNet(hw, hb), SetNet(aw, ab, av), device="cpu", log_time=False)
                pts = ref.TensorPoints(torch.tensor(start), padding_value=-1.0)
                pts.get_newton_polytope()
                for t in range(3):
                    obs, act, rew, done, nxt = game.step(pts, role, scale_observation=scale, exploration_rate=0.0)
                    if role == "host":
                        rec[tag + f"/t{t}_obs"], rec[tag + f"/t{t}_next"] = obs.numpy(), nxt.numpy()
                    else:
                        rec[tag + f"/t{t}_obs"], rec[tag + f"/t{t}_next"] = obs["points"].numpy(), nxt["points"].numpy()
                        rec[tag + f"/t{t}_obs_coords"] = obs["coords"].numpy()
                        rec[tag + f"/t{t}_next_coords"] = nxt["coords"].numpy()
                    rec[tag + f"/t{t}_actions"] = act.numpy().astype(np.int64)
                    rec[tag + f"/t{t}_rewards"] = rew.numpy()
                    rec[tag + f"/t{t}_dones"] = done.numpy()
                    rec[tag + f"/t{t}_state"] = pts.points.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "live_fused_game.npz"), **rec)
    return len(rec)


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not present: fixtures can only be regenerated next to the reference")
    ref = load_reference()
    which = sys.argv[1:] or ["torch", "list", "fused"]
    rng = np.random.default_rng(20260101)
    if "torch" in which or "list" in which:  # these two share one stream of draws
        n1 = make_torch(ref, rng)
        n2 = make_list(ref, rng)
        print(f"wrote live_torch.npz ({n1} arrays), live_list.npz ({n2} arrays)")
    if "fused" in which:
        n3 = make_fused_game(ref, np.random.default_rng(20260102))
        print(f"wrote live_fused_game.npz ({n3} arrays)")


if __name__ == "__main__":
    main()
