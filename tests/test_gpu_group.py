"""Device group of the C ABI (-m gpu): one process, N devices, one RCCL all-gather of the per-device summaries inside the
native library (ge_group_*, include/ge_step.h; SURVEY.md 8(e) process model).  A GPU box has one GPU, so the communicator
here has one rank; the host-side sum of the shards' own summaries stays in as the cross-check of the collective."""
import pytest

from conftest import load_dsl
from game_engine_amd import GameTable, GeError, RoomBatch, RoomGroup
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
