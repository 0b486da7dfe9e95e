"""Device group of the C ABI (-m gpu): one process, N devices, one RCCL all-gather of the per-device summaries inside the
native library (ge_group_*, include/ge_step.h; SURVEY.md 8(e) process model).  A GPU box has one GPU, so the communicator
here has one rank; the host-side sum of the shards' own summaries stays in as the cross-check of the collective."""
import pytest

from conftest import load_dsl
import numpy as np

from game_engine_amd import GameTable, GeError, RoomBatch, RoomGroup
from game_engine_amd.stepper import RoomShards, sum_summaries
from game_engine_amd.dist import reduce_summaries
from parity_util import assert_views_equal

pytestmark = pytest.mark.gpu


def _segs():
    ww, tt = GameTable(load_dsl("werewolf-(mafia)")), GameTable(load_dsl("two-truths-and-a-lie"))
    return [(ww, 8, 40000), (tt, 4, 30011)]


def test_one_device_group_equals_single_batch():
    seed, first = 0xC0FFEE, 1 << 33
    with RoomBatch(_segs(), seed=seed, first_room=first, restart=True) as b:
        b.step(70)
        want, want_rooms = b.summary(), b.read_rooms()
    with RoomGroup(_segs(), devices=[0], seed=seed, first_room=first, restart=True) as g:
        g.step(30)
        g.step(40)
        got = g.summary()                                   # per-device reduction + ncclAllGather + sum
        parts = g.shard_summaries()                         # the shards' own ge_batch_summary: host-side cross-check
        rooms = g.read_rooms()
    assert got == want
    assert len(parts) == 1 and parts[0] == want
    assert_views_equal(rooms, want_rooms, "group of one device vs one batch")


def test_duplicate_devices_and_bad_ordinals_are_refused_cleanly():
    for devices in ([0, 0], [0, 99], [-1], []):
        with pytest.raises(GeError) as e:
            RoomGroup(_segs(), devices=devices)
        assert e.value.status == -1, devices               # GE_ERR_ARG, before RCCL is asked (it forbids duplicate devices)
    ww = GameTable(load_dsl("werewolf-(mafia)"))
    with pytest.raises(GeError):
        RoomGroup([(ww, 8, 0)], devices=[0])


def test_group_summary_twice_and_after_more_steps():
    """The collective can be called repeatedly (progress reports every K turns) and follows later steps."""
    ww = GameTable(load_dsl("werewolf-(mafia)"))
    with RoomGroup([(ww, 12, 5000)], devices=[0], seed=3) as g, RoomBatch([(ww, 12, 5000)], seed=3) as b:
        for turns in (10, 25, 40):
            g.step(turns); b.step(turns)
            s1, s2 = g.summary(), g.summary()
            assert s1 == s2 == b.summary()


def _jobs():
    ww, tt = GameTable(load_dsl("werewolf-(mafia)")), GameTable(load_dsl("two-truths-and-a-lie"))
    return {"c4-shaped": [(ww, 12, 90001)],
            "c5-shaped": [(ww, 8, 50000), (tt, 4, 50000)],
            "four segments": [(ww, 8, 30001), (tt, 4, 777), (ww, 12, 4099), (tt, 6, 20000, 0b101)]}


@pytest.mark.parametrize("job", ["c4-shaped", "c5-shaped", "four segments"])
@pytest.mark.parametrize("n", [2, 3, 8])
def test_n_shards_on_one_device_equal_one_batch(job, n):
    """The native n > 1 sharding, executed without an n-GPU node: ge_group_partition's n shard descs created as n ordinary
    batches on device 0 (ge_batch_create_shard - what ge_group_create does per device, minus RCCL), stepped, and compared with ONE
    batch of the whole job: every room (through the reassembly RoomGroup.read_rooms uses), the shards' summaries summed as
    ge_group_summary sums them, and the global index of every part.  Fused and single-turn launches."""
    segs = _jobs()[job]
    seed, first = 0xC0FFEE, (1 << 35) + 12345
    for fuse, turns in ((0, (40, 33)), (1, (5, 4))):
        with RoomBatch(segs, seed=seed, first_room=first, restart=True, max_fuse=fuse) as b, \
                RoomShards(segs, devices=[0] * n, seed=seed, first_room=first, restart=True, max_fuse=fuse) as g:
            for t in turns:
                b.step(t); g.step(t)
            want, want_rooms = b.summary(), b.read_rooms()
            parts = g.shard_summaries()
            assert len(parts) == n and sum(p["rooms"] for p in parts) == want["rooms"]
            assert sum_summaries(parts) == want == g.summary(), (job, n, fuse)
            assert_views_equal(g.read_rooms(), want_rooms, f"{job}: {n} shards on one device vs one batch, fuse {fuse}")
            # the parts tile every segment: global firsts follow each other from the single batch's
            base = first
            for k, seg in enumerate(segs):
                nxt = base
                for i in range(n):
                    assert g.firsts[i][k] == nxt
                    nxt += seg[2] * (i + 1) // n - seg[2] * i // n
                assert nxt == base + seg[2]
                base += seg[2]


def test_a_shard_is_the_same_rooms_as_the_slice_of_the_whole_job():
    """One shard alone (part 1 of 3) holds exactly the rooms a slice of the single batch holds: nothing depends on its siblings."""
    from game_engine_amd.stepper import _job_desc, partition
    segs = _jobs()["c5-shaped"]
    with RoomBatch(segs, seed=9, first_room=77) as b:
        b.step(50)
        whole = b.read_rooms()
    with RoomShards(segs, devices=[0, 0, 0], seed=9, first_room=77) as g:
        g.step(50)
        d = _job_desc(segs, 9, 77, 0, False, False)
        sd, first = partition(d, 3, 1)
        lo0 = first[0] - 77
        lo1 = first[1] - 77
        one = g.read_rooms()
        assert_views_equal(one[lo0:lo0 + int(sd.seg[0].n_rooms)], whole[lo0:lo0 + int(sd.seg[0].n_rooms)], "segment 0 of part 1")
        assert_views_equal(one[lo1:lo1 + int(sd.seg[1].n_rooms)], whole[lo1:lo1 + int(sd.seg[1].n_rooms)], "segment 1 of part 1")
        assert np.array_equal(one, whole)
