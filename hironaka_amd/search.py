"""Batched Gumbel-MuZero search over HIP tree kernels -- the role `mctx.gumbel_muzero_policy` plays for
the reference (hironaka/jax/simulation_fn.py:85-117).

`mctx` is third-party and not available here, so this is the PUBLISHED algorithm (Danihelka et al.,
"Policy improvement by planning with Gumbel", ICLR 2022) with mctx's public names: a `Tree` with mctx's
fields (batch-first), `RootFnOutput`, `PolicyOutput(action, action_weights, search_tree)`,
`gumbel_muzero_policy(params, rng_key, root, recurrent_fn, num_simulations, ...)`.  Parity against mctx
itself is UNPINNED; the kernels are pinned to oracle/search_oracle.py (tests/test_gpu_search.py).

Per simulation the device runs: hk_search_select (one lane per ACTION walks the game's tree) -> gather of the parent embeddings -> `recurrent_fn` (opponent
policy + HIP environment step + the player's network) -> scatter of the new embeddings -> hk_search_backup; a recurrent_fn with an
`expander` (recurrent_fn.HostExpander / AgentExpander) runs gather / step / scatter as fused operators.  No host
synchronisation anywhere in the loop.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, NamedTuple, Optional, Tuple

import torch

from . import _abi as A
from ._lib import check, lib

UNVISITED = -1
ROOT_INDEX = 0


class RootFnOutput(NamedTuple):
    prior_logits: torch.Tensor  # [B, A]
    value: torch.Tensor  # [B]
    embedding: torch.Tensor  # [B, E]


class Tree(NamedTuple):
    node_visits: torch.Tensor  # [B, N] int32
    raw_values: torch.Tensor  # [B, N] f32
    node_values: torch.Tensor  # [B, N] f32
    parents: torch.Tensor  # [B, N] int32
    action_from_parent: torch.Tensor  # [B, N] int32
    children_index: torch.Tensor  # [B, N, A] int32
    children_prior_logits: torch.Tensor  # [B, N, A] f32
    children_visits: torch.Tensor  # [B, N, A] int32
    children_rewards: torch.Tensor  # [B, N, A] f32
    children_discounts: torch.Tensor  # [B, N, A] f32
    children_values: torch.Tensor  # [B, N, A] f32
    embeddings: torch.Tensor  # [B, N, E]
    root_invalid_actions: Optional[torch.Tensor]  # [B, A] uint8 or None

    @property
    def num_actions(self) -> int:
        return self.children_index.shape[-1]

    @property
    def num_simulations(self) -> int:
        return self.node_visits.shape[-1] - 1


class PolicyOutput(NamedTuple):
    action: torch.Tensor  # [B] int64
    action_weights: torch.Tensor  # [B, A] f32
    search_tree: Tree


def get_sequence_of_considered_visits(max_num_considered_actions: int, num_simulations: int) -> Tuple[int, ...]:
    """Sequential halving: for simulation i, the visit count the considered root actions must have."""
    if max_num_considered_actions <= 1:
        return tuple(range(num_simulations))
    log2max = int(math.ceil(math.log2(max_num_considered_actions)))
    sequence, visits = [], [0] * max_num_considered_actions
    num_considered = max_num_considered_actions
    while len(sequence) < num_simulations:
        num_extra_visits = max(1, int(num_simulations / (log2max * num_considered)))
        for _ in range(num_extra_visits):
            sequence.extend(visits[:num_considered])
            for i in range(num_considered):
                visits[i] += 1
        num_considered = max(2, num_considered // 2)
    return tuple(sequence[:num_simulations])


def get_table_of_considered_visits(max_num_considered_actions: int, num_simulations: int):
    return tuple(get_sequence_of_considered_visits(m, num_simulations)
                 for m in range(max_num_considered_actions + 1))


def simulation_key(rng_key, sim: int):
    """the key recurrent_fn receives in simulation `sim` (an int seed gets a different value per simulation)"""
    return (int(rng_key) + 1000003 * (sim + 1)) % (1 << 63) if isinstance(rng_key, int) else rng_key


def _tree_desc(tree: Tree) -> "A.hk_search_tree":
    t = A.hk_search_tree()
    for name in ("node_visits", "raw_values", "node_values", "parents", "action_from_parent", "children_index",
                 "children_prior_logits", "children_visits", "children_rewards", "children_discounts",
                 "children_values"):
        setattr(t, name, getattr(tree, name).data_ptr())
    t.batch, t.num_nodes, t.num_actions = tree.node_visits.shape[0], tree.node_visits.shape[1], tree.num_actions
    return t


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def instantiate_tree_from_root(root: RootFnOutput, num_simulations: int,
                               root_invalid_actions: Optional[torch.Tensor]) -> Tree:
    b, a = root.prior_logits.shape
    n = num_simulations + 1
    dev = root.prior_logits.device
    zi = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)
    zf = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    tree = Tree(zi(b, n), zf(b, n), zf(b, n), torch.full((b, n), -1, dtype=torch.int32, device=dev),
                torch.full((b, n), -1, dtype=torch.int32, device=dev),
                torch.full((b, n, a), UNVISITED, dtype=torch.int32, device=dev), zf(b, n, a), zi(b, n, a), zf(b, n, a),
                zf(b, n, a), zf(b, n, a),
                torch.zeros((b, n, root.embedding.shape[1]), dtype=root.embedding.dtype, device=dev),
                root_invalid_actions)
    tree.children_prior_logits[:, ROOT_INDEX] = root.prior_logits
    tree.raw_values[:, ROOT_INDEX] = root.value
    tree.node_values[:, ROOT_INDEX] = root.value
    tree.node_visits[:, ROOT_INDEX] = 1
    tree.embeddings[:, ROOT_INDEX] = root.embedding
    return tree


def _mask_invalid_actions(logits: torch.Tensor, invalid_actions: Optional[torch.Tensor]) -> torch.Tensor:
    if invalid_actions is None:
        return logits
    logits = logits - logits.max(dim=-1, keepdim=True).values
    return torch.where(invalid_actions.bool(), torch.full_like(logits, torch.finfo(logits.dtype).min), logits)


def _draw_gumbel(rng_key, shape, gumbel_scale: float, dev) -> torch.Tensor:
    gen = rng_key if isinstance(rng_key, torch.Generator) else torch.Generator(device=dev).manual_seed(int(rng_key))
    u = torch.rand(shape, generator=gen, device=dev, dtype=torch.float32).clamp_(min=1e-20, max=1.0 - 1e-7)
    return -torch.log(-torch.log(u)) * gumbel_scale


def _run_search(params, rng_key, root: RootFnOutput, gumbel: torch.Tensor, invalid_u8: Optional[torch.Tensor],
                table: torch.Tensor, recurrent_fn: Callable, num_simulations: int, max_depth: int,
                max_num_considered_actions: int) -> PolicyOutput:
    """the device work of one search (no host synchronisation, no host-to-device copies: capturable)"""
    b, a = root.prior_logits.shape
    dev = root.prior_logits.device
    tree = instantiate_tree_from_root(root, num_simulations, invalid_u8)
    desc = _tree_desc(tree)
    parent = torch.empty(b, dtype=torch.int32, device=dev)
    action = torch.empty(b, dtype=torch.int32, device=dev)
    node = torch.empty(b, dtype=torch.int32, device=dev)
    rows = torch.arange(b, device=dev)
    inv_ptr = None if invalid_u8 is None else invalid_u8.data_ptr()
    L = lib()
    # a recurrent_fn built by recurrent_fn.get_recurrent_fn_for_role may offer its expansion as fused operators
    # (recurrent_fn.HostExpander: gather / opponent argmax / step / features / scatter without the tensor-library glue)
    expander = getattr(recurrent_fn, "expander", None)
    if expander is not None and not expander.accepts(root.embedding):
        expander = None
    with torch.cuda.device(dev):
        expand_state = expander.begin(tree, root.embedding) if expander is not None else None
        if expand_state is not None and "tree" in expand_state:
            tree = expand_state["tree"]  # (an expander may keep the embeddings in its own layout: same fields, a view)
        for sim in range(num_simulations):
            check(L.hk_search_select(C.byref(desc), gumbel.data_ptr(), inv_ptr, table.data_ptr(),
                                     max_num_considered_actions, num_simulations, max_depth, sim + 1,
                                     parent.data_ptr(), action.data_ptr(), node.data_ptr(), _stream(gumbel)),
                  "hk_search_select")
            if expander is not None:
                step = expander.expand(params, simulation_key(rng_key, sim), tree, expand_state, parent, action, node)
            else:
                embedding = tree.embeddings[rows, parent]  # int32 indices are fine
                # (actions travel as int32, as in mctx; the HIP operators take them as they are)
                step, next_embedding = recurrent_fn(params, simulation_key(rng_key, sim), action, embedding)
                tree.embeddings[rows, node] = next_embedding.to(tree.embeddings.dtype)
            logits = step.prior_logits.to(torch.float32).contiguous()
            value = step.value.to(torch.float32).contiguous()
            reward = step.reward.to(torch.float32).contiguous()
            discount = step.discount.to(torch.float32).contiguous()
            check(L.hk_search_backup(C.byref(desc), parent.data_ptr(), action.data_ptr(), node.data_ptr(),
                                     logits.data_ptr(), value.data_ptr(), reward.data_ptr(), discount.data_ptr(),
                                     _stream(gumbel)), "hk_search_backup")
        final_action = torch.empty(b, dtype=torch.int32, device=dev)
        weights = torch.empty((b, a), dtype=torch.float32, device=dev)
        check(L.hk_search_policy(C.byref(desc), gumbel.data_ptr(), inv_ptr, final_action.data_ptr(),
                                 weights.data_ptr(), _stream(gumbel)), "hk_search_policy")
    return PolicyOutput(action=final_action.long(), action_weights=weights, search_tree=tree)


def _considered_visits_table(max_num_considered_actions: int, num_simulations: int, dev) -> torch.Tensor:
    return torch.tensor(get_table_of_considered_visits(max_num_considered_actions, num_simulations),
                        dtype=torch.int32, device=dev).reshape(max_num_considered_actions + 1, num_simulations)


def gumbel_muzero_policy(params, rng_key, root: RootFnOutput, recurrent_fn: Callable, num_simulations: int,
                         invalid_actions: Optional[torch.Tensor] = None, max_depth: Optional[int] = None, *,
                         max_num_considered_actions: int = 16, gumbel_scale: float = 1.0,
                         gumbel: Optional[torch.Tensor] = None) -> PolicyOutput:
    """rng_key: int seed or torch.Generator for the root Gumbel noise (or pass `gumbel` [B, A], already
    scaled).  recurrent_fn(params, rng_key, action [B] int64, embedding [B, E]) ->
    (RecurrentFnOutput(reward, discount, prior_logits, value), next_embedding)."""
    if not root.prior_logits.is_cuda:
        raise ValueError("the search runs on the HIP device: root tensors must be on the GPU")
    b, a = root.prior_logits.shape
    if a > 32:
        raise ValueError("at most 32 actions")
    dev = root.prior_logits.device
    max_depth = num_simulations if max_depth is None else max_depth
    invalid_u8 = None if invalid_actions is None else invalid_actions.to(torch.uint8).contiguous()
    root = RootFnOutput(_mask_invalid_actions(root.prior_logits.to(torch.float32), invalid_actions).contiguous(),
                        root.value.to(torch.float32).contiguous(), root.embedding.contiguous())
    if gumbel is None:
        gumbel = _draw_gumbel(rng_key, (b, a), gumbel_scale, dev)
    gumbel = gumbel.to(torch.float32).contiguous()
    table = _considered_visits_table(max_num_considered_actions, num_simulations, dev)
    return _run_search(params, rng_key, root, gumbel, invalid_u8, table, recurrent_fn, num_simulations, max_depth,
                       max_num_considered_actions)


class CapturedSearch:
    """One search of fixed shapes and fixed callables captured into a hipGraph: ~35 small launches per
    simulation (tree kernels, gathers, the opponent, the HIP step, the network) are replayed as one graph
    instead of being issued one by one from Python.

    Requirements on `recurrent_fn` (and what it calls): no host synchronisation, no host-to-device copies,
    random numbers only from torch's default generator (which graphs advance correctly) -- e.g. the fixed
    policies of `players.py` with `key=None`.  `params` are captured by reference: update weights in place.
    The returned PolicyOutput lives in the graph's memory and is overwritten by the next call."""

    def __init__(self, params, rng_key, root: RootFnOutput, recurrent_fn: Callable, num_simulations: int,
                 invalid_actions: Optional[torch.Tensor] = None, max_depth: Optional[int] = None, *,
                 max_num_considered_actions: int = 16, gumbel_scale: float = 1.0):
        b, a = root.prior_logits.shape
        dev = root.prior_logits.device
        self.gumbel_scale = gumbel_scale
        self.has_invalid = invalid_actions is not None
        self.logits = torch.zeros((b, a), dtype=torch.float32, device=dev)
        self.value = torch.zeros(b, dtype=torch.float32, device=dev)
        self.embedding = torch.zeros_like(root.embedding).contiguous()
        self.gumbel = torch.zeros((b, a), dtype=torch.float32, device=dev)
        self.invalid = torch.zeros((b, a), dtype=torch.uint8, device=dev) if self.has_invalid else None
        max_depth = num_simulations if max_depth is None else max_depth
        # everything the graph reads from outside its own memory pool must outlive it
        self.table = table = _considered_visits_table(max_num_considered_actions, num_simulations, dev)
        self.params, self.recurrent_fn = params, recurrent_fn
        self._fill(root, invalid_actions, rng_key)

        def run():
            return _run_search(params, None, RootFnOutput(self.logits, self.value, self.embedding), self.gumbel,
                               self.invalid, table, recurrent_fn, num_simulations, max_depth,
                               max_num_considered_actions)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            run()  # warm-up outside the capture (lazy initialisations, workspace allocations)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = run()

    def _fill(self, root: RootFnOutput, invalid_actions, rng_key):
        self.logits.copy_(_mask_invalid_actions(root.prior_logits.to(torch.float32), invalid_actions))
        self.value.copy_(root.value)
        self.embedding.copy_(root.embedding)
        self.gumbel.copy_(_draw_gumbel(rng_key, tuple(self.gumbel.shape), self.gumbel_scale, self.gumbel.device))
        if self.has_invalid:
            self.invalid.copy_(invalid_actions)

    def __call__(self, rng_key, root: RootFnOutput, invalid_actions: Optional[torch.Tensor] = None) -> PolicyOutput:
        if (invalid_actions is not None) != self.has_invalid:
            raise ValueError("this search was captured with" + ("" if self.has_invalid else "out") + " an action mask")
        self._fill(root, invalid_actions, rng_key)
        self.graph.replay()
        return self.out
