"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the batched Gumbel-MuZero search that the reference
obtains from the third-party package `mctx` (`mctx.gumbel_muzero_policy` with
`qtransform_completed_by_mix_value(use_mixed_value=True)`, called at hironaka/jax/simulation_fn.py:85-117).

PARITY UNPINNED: `mctx` is neither vendored in /root/reference nor installed here (setup.py:71 asks for
"mctx>=0.0.2", no lock file), and the reference's own tests only check weak invariants of the search
(jax/util.py:395-423).  This file restates the PUBLISHED algorithm -- Danihelka, Guez, Schrittwieser,
Silver, "Policy improvement by planning with Gumbel", ICLR 2022: Gumbel-top-k with sequential halving at
the root (sec. 3-4, appendix "Sequential Halving with Gumbel"), the deterministic interior selection
argmax(pi' - N/(1+sum N)) (sec. 5), completed Q-values with the mixed value estimate (appendix D) -- in the
form mctx's public release gives it (function and argument names kept so that a reader can line the two up).
It pins the HIP kernels (hironaka_amd/csrc/hk_search.h) to THIS restatement, nothing more.

Arithmetic contract shared with the kernels: tree statistics are float32 and the backward pass is plain
float32 (+, *, / in IEEE single, no contraction); everything that feeds an argmax or the final softmax
(softmax, completed Q-values, scores) is computed in float64 from the float32 statistics.
"""
from __future__ import annotations

import math
from typing import Callable, NamedTuple, Optional, Tuple

import numpy as np

UNVISITED = -1
ROOT = 0
F32_MIN = float(np.finfo(np.float32).min)
F32_TINY = float(np.finfo(np.float32).tiny)


def get_sequence_of_considered_visits(max_num_considered_actions: int, num_simulations: int) -> Tuple[int, ...]:
    """Sequential halving: for simulation i, the visit count the considered actions must have."""
    if max_num_considered_actions <= 1:
        return tuple(range(num_simulations))
    log2max = int(math.ceil(math.log2(max_num_considered_actions)))
    sequence = []
    visits = [0] * max_num_considered_actions
    num_considered = max_num_considered_actions
    while len(sequence) < num_simulations:
        num_extra_visits = max(1, int(num_simulations / (log2max * num_considered)))
        for _ in range(num_extra_visits):
            sequence.extend(visits[:num_considered])
            for i in range(num_considered):
                visits[i] += 1
        num_considered = max(2, num_considered // 2)
    return tuple(sequence[:num_simulations])


def get_table_of_considered_visits(max_num_considered_actions: int, num_simulations: int) -> np.ndarray:
    """[max_num_considered_actions + 1, num_simulations] int32: row m = the sequence for m considered actions."""
    return np.array([get_sequence_of_considered_visits(m, num_simulations)
                     for m in range(max_num_considered_actions + 1)], dtype=np.int32).reshape(
        max_num_considered_actions + 1, num_simulations)


class Tree(NamedTuple):
    node_visits: np.ndarray  # [B, N] int32
    raw_values: np.ndarray  # [B, N] f32
    node_values: np.ndarray  # [B, N] f32
    parents: np.ndarray  # [B, N] int32
    action_from_parent: np.ndarray  # [B, N] int32
    children_index: np.ndarray  # [B, N, A] int32
    children_prior_logits: np.ndarray  # [B, N, A] f32
    children_visits: np.ndarray  # [B, N, A] int32
    children_rewards: np.ndarray  # [B, N, A] f32
    children_discounts: np.ndarray  # [B, N, A] f32
    children_values: np.ndarray  # [B, N, A] f32
    embeddings: np.ndarray  # [B, N, E]


def new_tree(batch: int, num_nodes: int, num_actions: int, emb_dim: int, emb_dtype=np.float32) -> Tree:
    z = lambda *s, dt=np.float32: np.zeros(s, dtype=dt)
    return Tree(z(batch, num_nodes, dt=np.int32), z(batch, num_nodes), z(batch, num_nodes),
                np.full((batch, num_nodes), -1, np.int32), np.full((batch, num_nodes), -1, np.int32),
                np.full((batch, num_nodes, num_actions), UNVISITED, np.int32), z(batch, num_nodes, num_actions),
                z(batch, num_nodes, num_actions, dt=np.int32), z(batch, num_nodes, num_actions),
                z(batch, num_nodes, num_actions), z(batch, num_nodes, num_actions),
                z(batch, num_nodes, emb_dim, dt=emb_dtype))


def mask_invalid_actions(logits: np.ndarray, invalid: Optional[np.ndarray]) -> np.ndarray:
    """f32 in, f32 out: logits - max, invalid ones at the smallest float32."""
    if invalid is None:
        return logits.astype(np.float32)
    out = (logits - logits.max(axis=-1, keepdims=True)).astype(np.float32)
    return np.where(invalid.astype(bool), np.float32(F32_MIN), out).astype(np.float32)


def _seq_sum(x: np.ndarray) -> np.ndarray:
    """sum over the last axis in index order (numpy's own reduction is pairwise from 8 elements on; the
    kernels add left to right)"""
    acc = x[..., 0].copy()
    for a in range(1, x.shape[-1]):
        acc = acc + x[..., a]
    return acc


def _softmax64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float64)
    with np.errstate(invalid="ignore"):
        e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / _seq_sum(e)[..., None]


def completed_qvalues(tree: Tree, node: np.ndarray, value_scale=0.1, maxvisit_init=50.0, epsilon=1e-8) -> np.ndarray:
    """qtransform_completed_by_mix_value(use_mixed_value=True, rescale_values=True) at `node` [B] -> f64 [B, A]."""
    b = np.arange(node.shape[0])
    q = (tree.children_rewards[b, node].astype(np.float64)
         + tree.children_discounts[b, node].astype(np.float64) * tree.children_values[b, node].astype(np.float64))
    visits = tree.children_visits[b, node]
    raw = tree.raw_values[b, node].astype(np.float64)
    probs = np.maximum(F32_TINY, _softmax64(tree.children_prior_logits[b, node]))
    seen = visits > 0
    sum_visits = visits.sum(axis=-1).astype(np.float64)
    sum_probs = _seq_sum(np.where(seen, probs, 0.0))[..., None]
    weighted_q = _seq_sum(np.where(seen, probs * q / np.where(seen, sum_probs, 1.0), 0.0))
    value = (raw + sum_visits * weighted_q) / (sum_visits + 1.0)
    cq = np.where(seen, q, value[:, None])
    lo, hi = cq.min(axis=-1, keepdims=True), cq.max(axis=-1, keepdims=True)
    cq = (cq - lo) / np.maximum(hi - lo, epsilon)
    visit_scale = maxvisit_init + visits.max(axis=-1, keepdims=True).astype(np.float64)
    return visit_scale * value_scale * cq


def score_considered(considered_visit, gumbel, logits, cq, visits) -> np.ndarray:
    logits = logits.astype(np.float64) - logits.astype(np.float64).max(axis=-1, keepdims=True)
    penalty = np.where(visits == considered_visit[:, None], 0.0, -np.inf)
    return np.maximum(-1e9, gumbel.astype(np.float64) + logits + cq) + penalty


def root_action_selection(tree: Tree, gumbel, invalid, table, max_considered, b_idx) -> np.ndarray:
    node = np.zeros(len(b_idx), np.int64)
    visits = tree.children_visits[:, ROOT]
    cq = completed_qvalues(tree, node)
    num_valid = np.full(len(b_idx), visits.shape[1], np.int64)
    if invalid is not None:
        num_valid = num_valid - invalid.astype(np.int64).sum(-1)
    num_considered = np.minimum(max_considered, num_valid)
    considered_visit = table[num_considered, visits.sum(-1)]
    s = score_considered(considered_visit, gumbel, tree.children_prior_logits[:, ROOT], cq, visits)
    if invalid is not None:
        s = np.where(invalid.astype(bool), -np.inf, s)
    return s.argmax(axis=-1)


def interior_action_selection(tree: Tree, node: np.ndarray) -> np.ndarray:
    b = np.arange(node.shape[0])
    visits = tree.children_visits[b, node].astype(np.float64)
    cq = completed_qvalues(tree, node)
    probs = _softmax64(tree.children_prior_logits[b, node].astype(np.float64) + cq)
    return (probs - visits / (1.0 + visits.sum(-1, keepdims=True))).argmax(axis=-1)


def backward(tree: Tree, leaf: np.ndarray) -> None:
    f32 = np.float32
    for b in range(leaf.shape[0]):
        index = int(leaf[b])
        leaf_value = f32(tree.node_values[b, index])
        while index != ROOT:
            parent = int(tree.parents[b, index])
            count = f32(tree.node_visits[b, parent])
            action = int(tree.action_from_parent[b, index])
            leaf_value = f32(tree.children_rewards[b, parent, action]
                             + f32(tree.children_discounts[b, parent, action] * leaf_value))
            parent_value = f32(f32(f32(tree.node_values[b, parent] * count) + leaf_value) / f32(count + f32(1.0)))
            tree.children_values[b, parent, action] = tree.node_values[b, index]
            tree.children_visits[b, parent, action] += 1
            tree.node_values[b, parent] = parent_value
            tree.node_visits[b, parent] += 1
            index = parent


class PolicyOutput(NamedTuple):
    action: np.ndarray
    action_weights: np.ndarray
    search_tree: Tree


def gumbel_muzero_policy(params, root_prior_logits, root_value, root_embedding, recurrent_fn: Callable,
                         num_simulations: int, gumbel: np.ndarray, invalid_actions=None, max_depth=None,
                         max_num_considered_actions: int = 16, rng_key=None) -> PolicyOutput:
    """`gumbel` = gumbel_scale * Gumbel(0,1) noise [B, A], drawn by the caller (so that the kernels can be
    given the same numbers).  recurrent_fn(params, key, action [B] int, embedding [B, E]) ->
    ((reward, discount, prior_logits, value), next_embedding), all numpy; key = the per-simulation value
    hironaka_amd.search.simulation_key derives from an int `rng_key` (None otherwise)."""
    b, a = root_prior_logits.shape
    n = num_simulations + 1
    max_depth = num_simulations if max_depth is None else max_depth
    logits0 = mask_invalid_actions(np.asarray(root_prior_logits, np.float32), invalid_actions)
    tree = new_tree(b, n, a, root_embedding.shape[1], root_embedding.dtype)
    bi = np.arange(b)
    tree.children_prior_logits[:, ROOT] = logits0
    tree.raw_values[:, ROOT] = root_value
    tree.node_values[:, ROOT] = root_value
    tree.node_visits[:, ROOT] = 1
    tree.embeddings[:, ROOT] = root_embedding
    table = get_table_of_considered_visits(max_num_considered_actions, num_simulations)
    for sim in range(num_simulations):
        # -- simulate: walk down until an unvisited edge (or the depth limit)
        node = np.zeros(b, np.int64)
        action = root_action_selection(tree, gumbel, invalid_actions, table, max_num_considered_actions, bi)
        nxt = tree.children_index[bi, node, action].astype(np.int64)
        depth = np.zeros(b, np.int64)
        going = (nxt != UNVISITED) & (depth + 1 < max_depth)
        while going.any():
            node = np.where(going, nxt, node)
            depth = depth + going
            act_in = interior_action_selection(tree, node)
            action = np.where(going, act_in, action)
            nxt2 = tree.children_index[bi, node, action].astype(np.int64)
            nxt = np.where(going, nxt2, nxt)
            going = going & (nxt != UNVISITED) & (depth + 1 < max_depth)
        # -- expand
        new_node = np.where(nxt == UNVISITED, sim + 1, nxt)
        key = None if rng_key is None else (int(rng_key) + 1000003 * (sim + 1)) % (1 << 63)
        (reward, discount, prior_logits, value), emb = recurrent_fn(params, key, action, tree.embeddings[bi, node])
        tree.children_prior_logits[bi, new_node] = np.asarray(prior_logits, np.float32)
        tree.raw_values[bi, new_node] = value
        tree.node_values[bi, new_node] = value
        tree.node_visits[bi, new_node] += 1
        tree.embeddings[bi, new_node] = emb
        tree.children_index[bi, node, action] = new_node
        tree.children_rewards[bi, node, action] = reward
        tree.children_discounts[bi, node, action] = discount
        tree.parents[bi, new_node] = node
        tree.action_from_parent[bi, new_node] = action
        backward(tree, new_node)
    # -- the improved policy at the root
    visits = tree.children_visits[:, ROOT]
    cq = completed_qvalues(tree, np.zeros(b, np.int64))
    considered_visit = visits.max(axis=-1)
    s = score_considered(considered_visit, gumbel, tree.children_prior_logits[:, ROOT], cq, visits)
    if invalid_actions is not None:
        s = np.where(invalid_actions.astype(bool), -np.inf, s)
    action = s.argmax(axis=-1)
    search_logits = tree.children_prior_logits[:, ROOT].astype(np.float64) + cq
    if invalid_actions is not None:
        search_logits = search_logits - search_logits.max(axis=-1, keepdims=True)
        search_logits = np.where(invalid_actions.astype(bool), F32_MIN, search_logits)
    weights = _softmax64(search_logits).astype(np.float32)
    return PolicyOutput(action.astype(np.int32), weights, tree)


# ---- expansion glue (hironaka_amd/csrc/hk_search.h: expand_* / masked_argmax / mask_logits kernels) -------------------
# What hironaka/jax/recurrent_fn.py does between the search's select and backup, as index arithmetic: 84-104 for a
# host-role tree (class id -> subset, the agent observation, the agent's masked argmax), 105-121 for an agent-role tree
# (the host's argmax on the new points, the next agent observation); the agent's action mask is jax/util.py:287-305 in
# its NaN-free form.  `features` is a second per-node table next to the embeddings (hk_get_features of a node's points).
def _decode(cls: np.ndarray, dim: int) -> np.ndarray:
    from .np_oracle import decode_table
    ncls = 2 ** dim - dim - 1
    return decode_table(dim)[np.clip(np.asarray(cls), 0, ncls - 1)].astype(np.float32)


def _first_argmax_nan_wins(x: np.ndarray) -> np.ndarray:
    """argmax along the last axis; the first maximum; a NaN beats every number (the first NaN wins)"""
    x = np.asarray(x)
    nan = np.isnan(x)
    return np.where(nan.any(axis=-1), nan.argmax(axis=-1), np.where(nan, -np.inf, x).argmax(axis=-1)).astype(np.int32)


def expand_gather(embeddings, features, parent, action, dim):
    rows = np.arange(embeddings.shape[0])
    obs = embeddings[rows, parent]
    agent_feat = np.concatenate([features[rows, parent], _decode(action, dim)], axis=1)
    return obs, agent_feat


def masked_argmax(logits, action, dim):
    return _first_argmax_nan_wins(np.where(_decode(action, dim) > 0.5, logits, -np.inf))


def expand_scatter(obs, feat, node, embeddings, features):
    rows = np.arange(embeddings.shape[0])
    embeddings, features = embeddings.copy(), features.copy()
    embeddings[rows, node] = obs
    features[rows, node] = feat
    return embeddings, features


def expand_gather_agent(embeddings, parent, dim):
    rec = embeddings[np.arange(embeddings.shape[0]), parent]
    return rec[:, :-dim], rec[:, -dim:]


def expand_scatter_agent(points, feat, host_logits, node, embeddings, features, dim):
    rows = np.arange(embeddings.shape[0])
    cls = _first_argmax_nan_wins(host_logits)
    mask = _decode(cls, dim)
    embeddings, features = embeddings.copy(), features.copy()
    embeddings[rows, node] = np.concatenate([points, mask], axis=1)
    features[rows, node] = feat
    return embeddings, features, np.concatenate([feat, mask], axis=1), cls


def mask_logits(logits, class_id, dim):
    return np.where(_decode(class_id, dim) > 0.5, logits, -np.inf).astype(np.float32)
