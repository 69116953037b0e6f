"""Multi-GPU: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU
for tests), games sharded by rank, NO collective in the data path, one all-gather at the trainer
boundary -- the counterpart of the reference's leading device axis under ``pmap`` and its host-side
sums over that axis (hironaka/jax/jax_trainer.py:281-319, 513, 533-534).

The Philox counters carry the GLOBAL game index, so ``generate`` / ``rollout`` with
``game_offset = shard.start`` reproduce the unsharded batch bit for bit.
"""
from __future__ import annotations

from typing import Dict, NamedTuple, Optional, Sequence

import torch
import torch.distributed as dist


class Shard(NamedTuple):
    start: int  # global index of this rank's first game == game_offset
    size: int   # games on this rank
    total: int  # games over all ranks


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_range(total_games: int, rank_: Optional[int] = None, world_: Optional[int] = None) -> Shard:
    """contiguous, balanced split: the first (total % world) ranks own one game more"""
    r = rank() if rank_ is None else rank_
    w = world() if world_ is None else world_
    base, extra = divmod(total_games, w)
    size = base + (1 if r < extra else 0)
    start = r * base + min(r, extra)
    return Shard(start, size, total_games)


def _agree(ok: bool, device, what: str) -> None:
    """raise on EVERY rank if any rank found its input inconsistent (one tiny all-reduce): a rank that raised alone
    would leave the others waiting inside the collective that follows"""
    flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        raise ValueError(what if not ok else "another rank's local batch does not match its shard")


def all_gather_games(local: torch.Tensor, shard: Optional[Shard] = None, direct: bool = False,
                     rows_per_game: int = 1, validate: bool = True) -> torch.Tensor:
    """[B_local * rows_per_game, ...] per rank -> [B_total * rows_per_game, ...] on every rank, in global game order
    (rows_per_game > 1: a rollout flattened to one row per game and move).  Equal shards use one
    all_gather_into_tensor; ragged shards are padded to the largest one first.
    With a `shard` the local size is checked against it -- collectively (`validate`: one int32 all-reduce, a host
    synchronisation; pass validate=False inside timed or captured code).  Without a `shard` the ranks first exchange
    their local sizes (one tiny all-gather and a host synchronisation as well)."""
    w = world()
    if w == 1:
        return local
    k = int(rows_per_game)
    if direct:
        if shard is None:
            raise ValueError("the direct gather needs the Shard (every rank must know every shard's size)")
        return all_gather_games_direct(local, shard, rows_per_game=k, validate=validate)
    if shard is not None:
        sizes = [shard_range(shard.total, r, w).size * k for r in range(w)]
        ok = sizes[rank()] == local.shape[0]
        what = f"rank {rank()} holds {local.shape[0]} rows, its shard of {shard.total} games is {sizes[rank()]}"
        if validate:
            _agree(ok, local.device, what)
        elif not ok:
            raise ValueError(what)
    else:
        mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        every = torch.empty(w, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(every, mine)
        sizes = [int(v) for v in every.tolist()]
    if len(set(sizes)) == 1:
        out = torch.empty((w * local.shape[0], *local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    biggest = max(sizes)
    padded = torch.zeros((biggest, *local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((w * biggest, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * biggest: r * biggest + sizes[r]] for r in range(w)], dim=0)


def all_gather_games_direct(local: torch.Tensor, shard: Shard, rows_per_game: int = 1,
                            validate: bool = True) -> torch.Tensor:
    """The same gather as FULLY-CONNECTED point-to-point transfers: every rank posts one send of its shard to and one
    receive from every other rank in ONE batch (`dist.batch_isend_irecv` = a grouped send/recv in RCCL), i.e. the 7
    xGMI links of an MI355X carry 7 shards concurrently instead of a ring forwarding them hop by hop (SURVEY.md
    section 5 / 8(e): ~0.1 ms against ~0.7 ms for 15.7 MB per rank).  Ragged shards need no padding here.
    Opt-in (`all_gather_games(..., direct=True)`): RCCL's own all-gather is the default and the only variant that
    has run on hardware so far."""
    w, r = world(), rank()
    if w == 1:
        return local
    rpg = int(rows_per_game)
    sizes = [shard_range(shard.total, k, w).size * rpg for k in range(w)]
    starts = [shard_range(shard.total, k, w).start * rpg for k in range(w)]
    ok = sizes[r] == local.shape[0]
    what = f"rank {r} holds {local.shape[0]} rows, its shard of {shard.total} games is {sizes[r]}"
    if validate:
        _agree(ok, local.device, what)
    elif not ok:
        raise ValueError(what)
    local = local.contiguous()
    out = torch.empty((shard.total * rpg, *local.shape[1:]), dtype=local.dtype, device=local.device)
    out[starts[r]: starts[r] + sizes[r]] = local
    ops = []
    for k in range(w):
        if k == r or sizes[k] == 0:
            continue
        ops.append(dist.P2POp(dist.irecv, out[starts[k]: starts[k] + sizes[k]], k))
    for k in range(w):
        if k == r or sizes[r] == 0:
            continue
        ops.append(dist.P2POp(dist.isend, local, k))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def all_gather_rollout(rollout: Sequence[torch.Tensor], shard: Optional[Shard] = None, rows_per_game: int = 1):
    """(obs, policy, value) of each rank -> the full batch on every rank (trainer boundary).  `rows_per_game`:
    simulate()'s tensors are flattened to one row per game and move ([B * T, ...]: T rows per game)."""
    return tuple(all_gather_games(x, shard, rows_per_game=rows_per_game) for x in rollout)


def all_reduce_counts(counts: torch.Tensor) -> torch.Tensor:
    """per-step finished-game counts summed over the ranks (jax_trainer.py:533-534)"""
    if world() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts
