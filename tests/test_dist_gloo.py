"""The N>1 path on CPU: world_size 2 (and 3) over gloo.  The product has no CPU stepper, so the
ranks' per-shard summaries are produced here by the oracle (tests may use it) — what is under
test is game_engine_amd.dist: shard ranges keyed by global room index, the single all-gather,
and the reduction to the whole-job summary."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import load_dsl
from game_engine_amd import dist as gd
from game_engine_amd._lib import SUMMARY_WORDS


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_summary_words(game, n, seed, lo, hi, turns):
    """ge_summary words for rooms [lo, hi) computed from oracle rooms (checksum left 0)."""
    from oracle.oracle import Oracle
    orc = Oracle(load_dsl(game), n)
    rooms = orc.init_rooms(hi - lo)
    orc.run(rooms, seed, lo, 0, turns, threads=1)
    w = np.zeros(SUMMARY_WORDS, dtype=np.uint64)
    fin = rooms["end_turn"] >= 0
    alive = rooms["p"][:, :n, 2]
    wolves = ((rooms["p"][:, :n, 1] == 2) & (alive == 1)).sum(axis=1)
    w[0], w[1] = hi - lo, fin.sum()
    w[2], w[3] = (fin & (wolves == 0)).sum(), (fin & (wolves > 0)).sum()
    w[4], w[5] = alive.sum(), rooms["end_turn"][fin].sum()
    w[6:22] = np.bincount(np.minimum(rooms["end_turn"][fin] // 8, 15), minlength=16)
    w[39] = turns
    return w


def _worker(rank, world, port, total, turns, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = gd.shard_range(total, world, rank)
    local = _oracle_summary_words("werewolf-(mafia)", 8, 7, lo, hi, turns)
    gathered = gd.allgather_summary_words(local, world)
    q.put((rank, lo, hi, gd.reduce_summaries(gathered).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_summary_equals_whole_job(world):
    total, turns = 3001, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, turns, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    whole = _oracle_summary_words("werewolf-(mafia)", 8, 7, 0, total, turns).tolist()
    ranges = sorted((lo, hi) for _, lo, hi, _ in results)
    assert ranges[0][0] == 0 and ranges[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    for _, _, _, summed in results:           # every rank ends up with the whole-job summary
        assert summed == whole


def test_shard_ranges_cover_exactly():
    for total in (1, 7, 65536, 16777216, 1000003):
        for world in (1, 2, 3, 4, 8):
            edges = [gd.shard_range(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    assert gd.shard_first_room(65536, 3) == 196608


def test_reduce_wraps_like_the_device():
    w = np.zeros((2, SUMMARY_WORDS), dtype=np.uint64)
    w[:, 38] = np.uint64(2**63 + 5)             # checksum words wrap mod 2^64
    w[:, 39] = 64
    out = gd.reduce_summaries(w)
    assert int(out[38]) == 10 and int(out[39]) == 64
