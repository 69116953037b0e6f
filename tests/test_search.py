"""Gumbel-MuZero search.  The reference takes it from the third-party package mctx,
which is not available: PARITY UNPINNED.  CPU tests: the sequential-halving schedule and invariants of
the restatement in oracle/search_oracle.py.  GPU tests: the HIP tree kernels + driver against that
restatement, on the real environment."""
import numpy as np
import pytest

from oracle import search_oracle as SO


def test_sequential_halving_schedule():
    # worked by hand from the definition: 4 considered actions, 16 simulations -> two rounds over 4 actions,
    # then 4 rounds over the best 2
    assert SO.get_sequence_of_considered_visits(4, 16) == (0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5)
    assert SO.get_sequence_of_considered_visits(1, 5) == (0, 1, 2, 3, 4)
    assert SO.get_sequence_of_considered_visits(0, 3) == (0, 1, 2)
    for m in (2, 3, 4, 11, 16):
        for n in (1, 7, 32, 50):
            seq = SO.get_sequence_of_considered_visits(m, n)
            assert len(seq) == n and seq[0] == 0
            assert all(b - a in (0, 1) or b < a for a, b in zip(seq, seq[1:]))
    table = SO.get_table_of_considered_visits(4, 32)
    assert table.shape == (5, 32) and table.dtype == np.int32


def _toy_recurrent_fn(num_actions, seed):
    """a deterministic synthetic environment + evaluator on integer embeddings"""
    rng = np.random.default_rng(seed)
    w = rng.normal(size=(7, num_actions)).astype(np.float32)

    def recurrent_fn(params, key, action, embedding):
        nxt = (embedding * 3 + action[:, None] + 1) % 7
        feat = np.eye(7, dtype=np.float32)[nxt[:, 0].astype(np.int64)]
        logits = feat @ w
        value = np.tanh(feat @ w[:, 0]).astype(np.float32)
        reward = ((nxt[:, 0] == 0) * 1.0).astype(np.float32)
        discount = np.full(len(action), 0.99, np.float32)
        return (reward, discount, logits, value), nxt.astype(embedding.dtype)

    return recurrent_fn


@pytest.mark.parametrize("num_actions,num_simulations,max_considered", [(4, 32, 16), (3, 8, 2), (11, 20, 4), (4, 5, 16)])
def test_oracle_search_invariants(num_actions, num_simulations, max_considered):
    rng = np.random.default_rng(5)
    b = 64
    logits = rng.normal(size=(b, num_actions)).astype(np.float32)
    value = rng.normal(size=b).astype(np.float32)
    emb = rng.integers(0, 7, size=(b, 1)).astype(np.float32)
    gumbel = (0.3 * rng.gumbel(size=(b, num_actions))).astype(np.float32)
    out = SO.gumbel_muzero_policy((), logits, value, emb, _toy_recurrent_fn(num_actions, 1), num_simulations, gumbel,
                                  max_num_considered_actions=max_considered)
    t = out.search_tree
    assert np.all(t.children_visits[:, 0].sum(-1) == num_simulations)
    assert np.all(t.node_visits[:, 0] == num_simulations + 1)
    assert np.allclose(out.action_weights.sum(-1), 1.0, atol=1e-6) and np.all(out.action_weights >= 0)
    rows = np.arange(b)
    assert np.all(t.children_visits[rows, 0, out.action] == t.children_visits[:, 0].max(-1))
    assert np.all(t.children_index[rows, 0, out.action] >= 1)
    # every expanded node hangs under the edge that points to it
    for g in range(b):
        for n in range(1, num_simulations + 1):
            p, a = t.parents[g, n], t.action_from_parent[g, n]
            if p >= 0:
                assert t.children_index[g, p, a] == n
    # the number of root actions ever tried is bounded by the Gumbel-top-k size
    assert np.all((t.children_visits[:, 0] > 0).sum(-1) <= min(max_considered, num_actions))
    # with an invalid-action mask the masked actions are never visited nor chosen
    invalid = np.zeros((b, num_actions), np.uint8)
    invalid[:, 0] = 1
    out2 = SO.gumbel_muzero_policy((), logits, value, emb, _toy_recurrent_fn(num_actions, 1), num_simulations, gumbel,
                                   invalid_actions=invalid, max_num_considered_actions=max_considered)
    assert np.all(out2.search_tree.children_visits[:, 0, 0] == 0) and np.all(out2.action != 0)
    assert np.all(out2.action_weights[:, 0] < 1e-12)


def test_oracle_search_depth_limit():
    rng = np.random.default_rng(6)
    b, a, n = 32, 4, 24
    logits = rng.normal(size=(b, a)).astype(np.float32)
    out = SO.gumbel_muzero_policy((), logits, np.zeros(b, np.float32), np.zeros((b, 1), np.float32),
                                  _toy_recurrent_fn(a, 2), n, np.zeros((b, a), np.float32), max_depth=2,
                                  max_num_considered_actions=2)
    t = out.search_tree
    # depth of every node <= 2
    for g in range(b):
        for node in range(1, n + 1):
            if t.parents[g, node] >= 0:
                d, x = 0, node
                while x != 0:
                    x = t.parents[g, x]
                    d += 1
                assert d <= 2
    assert np.all(t.node_visits[:, 0] == n + 1)


def test_expansion_glue_restatement():
    """oracle/search_oracle.py's expansion glue: the argmax rule equals the tensor library's (first maximum, NaN wins),
    decode clamps, gather / scatter are inverse on the touched rows"""
    import torch
    rng = np.random.default_rng(0)
    x = rng.standard_normal((200, 4)).astype(np.float32)
    x[rng.random(x.shape) < 0.15] = np.nan
    x[rng.random(x.shape) < 0.15] = 0.5
    assert np.array_equal(SO._first_argmax_nan_wins(x), torch.argmax(torch.tensor(x), dim=1).numpy())
    b, n, m, d = 50, 6, 5, 3
    e = m * d
    emb = rng.standard_normal((b, n, e)).astype(np.float32)
    feat = rng.standard_normal((b, n, e)).astype(np.float32)
    parent = rng.integers(0, n, b)
    action = rng.integers(-2, 7, b)
    obs, af = SO.expand_gather(emb, feat, parent, action, d)
    assert obs.shape == (b, e) and af.shape == (b, e + d)
    assert set(np.unique(af[:, e:])) <= {0.0, 1.0} and (af[:, e:].sum(axis=1) >= 2).all()
    emb2, feat2 = SO.expand_scatter(obs, af[:, :e], parent, emb, feat)
    assert np.array_equal(emb2, emb) and np.array_equal(feat2, feat)
    ax = SO.masked_argmax(x[:b, :d], action, d)
    assert (af[np.arange(b), e + ax] == 1.0).all()
    ml = SO.mask_logits(x[:b, :d], np.clip(action, 0, 3), d)
    assert np.isneginf(ml[af[:, e:] == 0.0]).all()
