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


def all_gather_rollout(rollout: Sequence[torch.Tensor], shard: Optional[Shard] = None, rows_per_game: int = 1,
                       validate: bool = True):
    """(obs, policy, value) of each rank -> the full batch on every rank (trainer boundary).  `rows_per_game`:
    simulate()'s tensors are flattened to one row per game and move ([B * T, ...]: T rows per game).
    The local sizes are checked ONCE, collectively, on the first tensor (one int32 all-reduce + a host synchronisation,
    `validate`); the tensors of one rollout share their leading dimension, which is asserted locally."""
    rollout = tuple(rollout)
    if not rollout:
        return rollout
    rows = rollout[0].shape[0]
    if any(x.shape[0] != rows for x in rollout):
        raise ValueError("the tensors of a rollout share their leading dimension")
    first = all_gather_games(rollout[0], shard, rows_per_game=rows_per_game, validate=validate)
    if shard is None:  # (the sizes were exchanged for the first tensor: a Shard-less gather of the rest exchanges them again)
        return (first,) + tuple(all_gather_games(x, None, rows_per_game=rows_per_game) for x in rollout[1:])
    return (first,) + tuple(all_gather_games(x, shard, rows_per_game=rows_per_game, validate=False)
                            for x in rollout[1:])


class GatherPipeline:
    """The trainer-boundary gather BEHIND the next episode.  Episodes are independent, so the all-gather of episode i's
    final states need not sit between episode i and episode i + 1 on the launch stream (where its latency is the whole
    N > 1 curve): `submit` hands the states to a side stream behind an event and returns at once; the producer runs
    episode i + 1 into its OTHER buffer meanwhile; `result` makes the current stream wait for a gather.  The reference's
    counterpart is the host-side use of per-device results after the pmapped loop (jax_trainer.py:513,533-534).

    Double-buffered by the CALLER: a tensor handed to `submit` must not be overwritten before `depth` further submits
    (bench.py alternates two state buffers; `copy=True` stages a private copy instead: one device copy per gather).
    Backend nccl (RCCL): the collective is enqueued on the side stream.  Host tensors / gloo (the CPU rehearsal): the
    gather runs at `submit` (there is no stream to overlap with) -- the same results, the same protocol."""

    def __init__(self, shard: Shard, rows_per_game: int = 1, depth: int = 2, direct: bool = False):
        self.shard, self.rows_per_game, self.depth, self.direct = shard, int(rows_per_game), int(depth), bool(direct)
        self._out = [None] * self.depth
        self._done = [None] * self.depth
        self._stage = [None] * self.depth
        self._side = None
        self._n = 0

    def submit(self, local: torch.Tensor, copy: bool = False) -> int:
        slot = self._n % self.depth
        self._n += 1
        if not local.is_cuda:
            self._out[slot] = all_gather_games(local, self.shard, direct=self.direct, rows_per_game=self.rows_per_game,
                                               validate=False)
            self._done[slot] = None
            return self._n - 1
        cur = torch.cuda.current_stream(local.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=local.device)
        if copy:
            if self._stage[slot] is None or self._stage[slot].shape != local.shape:
                self._stage[slot] = torch.empty_like(local)
            if self._done[slot] is not None:
                cur.wait_event(self._done[slot])  # the slot's previous gather has read its staging buffer
            self._stage[slot].copy_(local)
            local = self._stage[slot]
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            out = all_gather_games(local, self.shard, direct=self.direct, rows_per_game=self.rows_per_game, validate=False)
            done = torch.cuda.Event()
            done.record(self._side)
        self._out[slot], self._done[slot] = out, done
        return self._n - 1

    def result(self, ticket: Optional[int] = None) -> torch.Tensor:
        """the gathered tensor of `ticket` (default: the latest submit); the current stream waits for it.  Valid until
        `depth` further submits."""
        ticket = self._n - 1 if ticket is None else ticket
        if ticket < 0 or ticket < self._n - self.depth or ticket >= self._n:
            raise ValueError(f"ticket {ticket} is not among the last {self.depth} submits")
        slot = ticket % self.depth
        if self._done[slot] is not None:
            torch.cuda.current_stream(self._out[slot].device).wait_event(self._done[slot])
        return self._out[slot]

    def wait(self, ticket: Optional[int]) -> None:
        """the current stream waits until the gather of `ticket` has READ its input (before the producer overwrites that
        buffer).  A ticket older than `depth` submits is covered by the later gather of its slot (the side stream runs
        them in order)."""
        if ticket is None or ticket < 0 or ticket >= self._n:
            return
        slot = ticket % self.depth
        if self._done[slot] is not None and self._out[slot] is not None:
            torch.cuda.current_stream(self._out[slot].device).wait_event(self._done[slot])

    def drain(self) -> None:
        """every submitted gather is finished when the current stream passes this point"""
        for slot in range(self.depth):
            if self._done[slot] is not None and self._out[slot] is not None:
                torch.cuda.current_stream(self._out[slot].device).wait_event(self._done[slot])


def all_reduce_counts(counts: torch.Tensor) -> torch.Tensor:
    """per-step finished-game counts summed over the ranks (jax_trainer.py:533-534)"""
    if world() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts
