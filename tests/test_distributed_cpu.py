"""World-size-2 tests of the multi-GPU path on CPU (gloo): shard arithmetic, the all-gather at the
trainer boundary and the sharding invariance of the rollout (each rank computes its shard with
game_offset = shard.start -- here through the oracle, standing in for the kernels that need a GPU --
and the gathered result must equal the unsharded run)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hironaka_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import c_oracle as CO
        CO.set_threads(1)
        shard = D.shard_range(total)
        assert D.world() == world and D.rank() == rank
        m, d, T = 10, 3, 6
        local = CO.generate_points(shard.size, m, d, 20, 42, game_offset=shard.start)
        final, rec = CO.rollout(local, T, 7, game_offset=shard.start)
        gathered = D.all_gather_games(torch.from_numpy(final), shard)
        # without a Shard the ranks exchange their sizes first: ragged inputs take the padded path too
        assert torch.equal(D.all_gather_games(torch.from_numpy(final)), gathered)
        # ... and the fully-connected point-to-point variant gives the same tensor
        assert torch.equal(D.all_gather_games(torch.from_numpy(final), shard, direct=True), gathered)
        try:
            D.all_gather_games(torch.from_numpy(final)[:-1], shard)
            raise AssertionError("a local batch that is not the rank's shard must be rejected")
        except ValueError:
            pass
        obs = D.all_gather_rollout((torch.from_numpy(rec["obs"][0]), torch.from_numpy(rec["axis"][0]),
                                    torch.from_numpy(rec["reward"][0])), shard)
        # simulate()'s tensors are flattened to [B * T, ...] rows: T rows per game, gathered in game order
        flat = torch.from_numpy(np.ascontiguousarray(rec["obs"].transpose(1, 0, 2, 3)).reshape(shard.size * T, m * d))
        rows = D.all_gather_rollout((flat, flat[:, :1].clone()), shard, rows_per_game=T)
        assert rows[0].shape[0] == total * T and rows[1].shape == (total * T, 1)
        assert torch.equal(D.all_gather_games(flat, shard, direct=True, rows_per_game=T), rows[0])
        try:  # every rank raises together (a rank raising alone would leave the others inside the collective)
            D.all_gather_games(flat[: flat.shape[0] - (1 if rank == 0 else 0)], shard, rows_per_game=T)
            raise AssertionError("an inconsistent local batch on ONE rank must raise on every rank")
        except ValueError:
            pass
        # the pipelined gather (bench.py N > 1, HipTrainer.simulate): tickets, double buffering, the serial gather's result
        pipe = D.GatherPipeline(shard, depth=2)
        finals = [torch.from_numpy(CO.rollout(local, T, 7 + e, game_offset=shard.start, record=False)[0]) for e in range(3)]
        tickets = [pipe.submit(f) for f in finals[:2]]
        assert torch.equal(pipe.result(tickets[0]), D.all_gather_games(finals[0], shard))
        tickets.append(pipe.submit(finals[2]))
        assert torch.equal(pipe.result(), D.all_gather_games(finals[2], shard))
        assert torch.equal(pipe.result(tickets[1]), D.all_gather_games(finals[1], shard))
        try:
            pipe.result(tickets[0])
            raise AssertionError("a ticket older than the pipeline's depth is gone")
        except ValueError:
            pass
        pipe.drain()
        counts = D.all_reduce_counts(torch.from_numpy(rec["done_count"].astype(np.int64)))
        if rank == 0:
            np.savez(os.path.join(tmpdir, "out.npz"), final=gathered.numpy(), obs=obs[0].numpy(), axis=obs[1].numpy(),
                     reward=obs[2].numpy(), counts=counts.numpy(), rows=rows[0].numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,world", [(64, 2), (37, 2), (37, 3)])  # equal and ragged shards
def test_two_rank_shards_equal_unsharded(tmp_path, total, world):
    from oracle import c_oracle as CO
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    full = CO.generate_points(total, 10, 3, 20, 42)
    final, rec = CO.rollout(full, 6, 7)
    assert np.array_equal(got["final"], final)
    assert np.array_equal(got["obs"], rec["obs"][0]) and np.array_equal(got["axis"], rec["axis"][0])
    assert np.array_equal(got["reward"], rec["reward"][0])
    assert np.array_equal(got["counts"], rec["done_count"].astype(np.int64))
    assert np.array_equal(got["rows"], np.ascontiguousarray(rec["obs"].transpose(1, 0, 2, 3)).reshape(total * 6, 30))


def test_shard_range_partitions():
    for total in (0, 1, 7, 64, 65536, 524288 + 3):
        for world in (1, 2, 3, 8):
            shards = [D.shard_range(total, r, world) for r in range(world)]
            assert shards[0].start == 0 and sum(s.size for s in shards) == total
            for a, b in zip(shards, shards[1:]):
                assert b.start == a.start + a.size and a.size - b.size in (0, 1)
    assert D.world() == 1 and D.rank() == 0
    x = torch.arange(6).reshape(3, 2)
    assert D.all_gather_games(x) is x
