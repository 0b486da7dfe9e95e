"""The N>1 path with the PRODUCT on a GPU (-m gpu): two ranks (both on GPU 0 — the test box has one
card; collectives over gloo, as bench.py's GE_DIST_BACKEND=gloo rehearsal does) each step their shard
of the global room range; after the single all-gather every rank holds the whole-job summary, equal
— checksum included — to the summary of one batch over all rooms."""
import os
import socket

import pytest
import torch.multiprocessing as mp

from conftest import load_dsl

pytestmark = pytest.mark.gpu
TOTAL, TURNS, SEED, FIRST = 50001, 64, 0xC0FFEE, 1 << 33


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from game_engine_amd import GameTable, RoomBatch
    from game_engine_amd import dist as gd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = gd.shard_range(TOTAL, world, rank)
    tb = GameTable(load_dsl("werewolf-(mafia)"))
    with RoomBatch([(tb, 8, hi - lo)], seed=SEED, first_room=FIRST + lo, device=0) as b:
        b.step(TURNS)
        s = gd.allgather_summary(b, world)
    q.put((rank, s))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_summary_equals_single_batch():
    from game_engine_amd import GameTable, RoomBatch
    with RoomBatch([(GameTable(load_dsl("werewolf-(mafia)")), 8, TOTAL)], seed=SEED, first_room=FIRST) as b:
        b.step(TURNS)
        whole = b.summary()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, s in results:
        assert s == whole
