"""C-ABI behaviour beyond plain stepping (-m gpu): turn restore, stream changes, device-side reset,
batched injection of host-driven players' actions, graph cache eviction."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_dsl
from game_engine_amd import GameTable, GeError, RoomBatch
from oracle.oracle import Oracle
from parity_util import assert_views_equal, oracle_rooms_as_views

pytestmark = pytest.mark.gpu


def _hip():
    return C.CDLL("libamdhip64.so")


def test_set_turn_restores_raw_state_checkpoint(dsl_ww):
    """Checkpoint = D2H copy of ge_batch_state + the turn; restore into a fresh batch by copying it
    back and ge_batch_set_turn: bit-identical continuation, with and without the hipGraph path."""
    tb = GameTable(dsl_ww)
    for fuse in (64, 1):
        with RoomBatch([(tb, 8, 5000)], seed=9, first_room=77, max_fuse=fuse, restart=True) as a:
            a.step(37)
            a.sync()
            ptr, nbytes, _ = a.state(0)
            ck = np.empty(nbytes, dtype=np.uint8)
            assert _hip().hipMemcpy(C.c_void_p(ck.ctypes.data), C.c_void_p(ptr), C.c_size_t(nbytes), 2) == 0     # D2H
            a.step(50)
            end = a.read_rooms().tobytes()
        with RoomBatch([(tb, 8, 5000)], seed=9, first_room=77, max_fuse=fuse, restart=True) as b:
            ptr, nbytes, _ = b.state(0)
            assert _hip().hipMemcpy(C.c_void_p(ptr), C.c_void_p(ck.ctypes.data), C.c_size_t(nbytes), 1) == 0      # H2D
            b.set_turn(37)
            b.step(50)
            assert b.turn == 87 and b.read_rooms().tobytes() == end
    with RoomBatch([(tb, 8, 10)], seed=1) as b:
        with pytest.raises(GeError) as e:
            b.set_turn(1 << 32)
        assert e.value.status == -6


def test_raw_restore_under_another_seed_drops_the_prepared_deals(dsl_ww):
    """Werewolf x 8 records carry a prepared role deal (a cache keyed by seed, global room and game index) in their spare
    half-word.  Raw planes copied into a batch with ANOTHER seed and room range must not deal from it: ge_batch_set_turn, the
    call that completes a raw restore, drops the caches.  The continuation equals the oracle's from the same state."""
    from parity_util import views_as_oracle_rooms
    tb = GameTable(dsl_ww)
    R = 6000
    with RoomBatch([(tb, 8, R)], seed=9, first_room=77, max_fuse=1, restart=True) as a:
        a.step(48)                                           # single-turn launches: deals are prepared into the records
        a.sync()
        views = a.read_rooms()
        ptr, nbytes, _ = a.state(0)
        raw = np.empty(nbytes, dtype=np.uint8)
        assert _hip().hipMemcpy(C.c_void_p(raw.ctypes.data), C.c_void_p(ptr), C.c_size_t(nbytes), 2) == 0
    plane1 = raw.view(np.uint32).reshape(2, -1, 4)[1]        # words 4..7 of every (padded) room
    assert (plane1[:R, 3] >> 16).any()                       # some room does carry a prepared deal
    orc = Oracle(dsl_ww, 8)
    for fuse in (1, 64):
        with RoomBatch([(tb, 8, R)], seed=1234, first_room=5, max_fuse=fuse, restart=True) as b:
            ptr, nbytes, _ = b.state(0)
            assert _hip().hipMemcpy(C.c_void_p(ptr), C.c_void_p(raw.ctypes.data), C.c_size_t(nbytes), 1) == 0
            b.set_turn(48)
            b.step(60)
            rooms = views_as_oracle_rooms(orc, views)
            orc.run(rooms, 1234, 5, 48, 60, threads=0, restart=True)
            assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"raw restore under another seed, fuse {fuse}")


def test_steps_on_alternating_streams_are_ordered(dsl_ww):
    """Consecutive ge_batch_step calls on different streams: the library orders them with an event, and
    reads wait for all of it."""
    tb = GameTable(dsl_ww)
    with RoomBatch([(tb, 8, 200000)], seed=4, max_fuse=8, restart=True) as a:
        a.step(96)
        want = a.read_rooms().tobytes()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with RoomBatch([(tb, 8, 200000)], seed=4, max_fuse=8, restart=True) as b:
        for k in range(12):
            b.step(8, (s1, s2)[k % 2].cuda_stream)
        assert b.read_rooms().tobytes() == want


def test_reset_is_a_device_fill_back_to_the_template(dsl_ww, dsl_tt):
    for dsl, n in ((dsl_ww, 12), (dsl_tt, 4)):
        tb = GameTable(dsl)
        with RoomBatch([(tb, n, 70001)], seed=2) as b:
            fresh = b.read_rooms().tobytes()
            b.step(40)
            first = b.read_rooms().tobytes()
            assert first != fresh
            b.reset()
            assert b.turn == 0 and b.read_rooms().tobytes() == fresh
            b.step(40)
            assert b.read_rooms().tobytes() == first


def test_current_device_is_left_alone(dsl_ww):
    before = torch.cuda.current_device()
    with RoomBatch([(GameTable(dsl_ww), 8, 100)], seed=2, device=0) as b:
        b.step(3)
        b.summary()
    assert torch.cuda.current_device() == before


def test_graph_cache_eviction_while_running(dsl_ww):
    """More distinct n_turns than the graph cache holds, back to back without synchronising: an evicted
    executable must not be destroyed while it may still run."""
    tb = GameTable(dsl_ww)
    seq = [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 4, 14, 5]
    with RoomBatch([(tb, 8, 300000)], seed=6, max_fuse=1, restart=True) as b:
        for k in seq:
            b.step(k)
        got = b.read_rooms().tobytes()
    with RoomBatch([(tb, 8, 300000)], seed=6, max_fuse=64, restart=True) as a:
        a.step(sum(seq))
        assert a.read_rooms().tobytes() == got


@pytest.mark.parametrize("game,n,mask", [("werewolf-(mafia)", 8, 0b101), ("werewolf-(mafia)", 12, 0b100000000001),
                                         ("two-truths-and-a-lie", 4, 0b11), ("two-truths-and-a-lie", 9, 0b100000001)])
def test_batched_injection_equals_oracle(game, n, mask):
    """ge_batch_inject_actions: thousands of host-driven players' actions per call (several per room,
    valid and invalid mixed, rooms in random order) against the oracle applying them one by one."""
    dsl = load_dsl(game)
    orc = Oracle(dsl, n)
    R, seed, first = 20000, 33, 1 << 20
    rng = np.random.default_rng(n * 7 + mask)
    rooms = orc.init_rooms(R)
    humans = [i + 1 for i in range(n) if (mask >> i) & 1]
    applied = refused = 0
    with RoomBatch([(GameTable(dsl), n, R, mask)], seed=seed, first_room=first, max_fuse=1) as b:
        for t in range(60):
            k = 6000
            rr = rng.integers(0, R, size=k).astype(np.uint64)
            pl = rng.choice(humans, size=k).astype(np.uint32)
            ch = rng.integers(0, n + 2, size=k).astype(np.uint32)
            if t % 7 == 0:
                rr[:3] = [R, R + 5, 1 << 40]                  # outside the batch
            want = np.array([0 if (r < R and orc.inject(rooms, int(r), int(p), int(c))) else (-6 if r >= R else -1)
                             for r, p, c in zip(rr, pl, ch)], dtype=np.int32)
            got = b.inject_actions(rr, pl, ch)
            assert got.tolist() == want.tolist(), f"{game} turn {t}"
            applied += int((got == 0).sum()); refused += int((got != 0).sum())
            b.step(1)
            orc.run(rooms, seed, first, t, 1, threads=0, human_mask=mask)
            assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, rooms), f"{game} turn {t}")
        # return value = status of the first refused action in input order; NULL status is allowed
        lib, h = b._lib, b._h
        r3 = (C.c_uint64 * 3)(0, R + 1, 1)
        p3 = (C.c_uint32 * 3)(99, 1, 99)
        c3 = (C.c_uint32 * 3)(1, 1, 1)
        assert lib.ge_batch_inject_actions(h, 3, r3, p3, c3, None) == -1
        assert lib.ge_batch_inject_actions(h, 2, C.byref(r3, 8), C.byref(p3, 4), C.byref(c3, 4), None) == -6
        assert lib.ge_batch_inject_actions(h, 0, None, None, None, None) == 0
        assert lib.ge_batch_inject_actions(h, 1, None, p3, c3, None) == -1
    assert applied > 1000 and refused > 1000


def test_room_views_through_the_threaded_host_conversion(dsl_ww, dsl_tt, monkeypatch):
    """ge_batch_read_rooms / write_rooms convert between packed planes and canonical views on several host threads from 16 K
    rooms on (GE_IO_THREADS caps them).  A mixed batch large enough for that: one thread and many give the same bytes, sub-ranges
    across the segment boundary equal slices of the whole, a reused `out` array is overwritten completely, write -> read is the
    identity, and a write with one bad view anywhere changes nothing."""
    segs = [(GameTable(dsl_ww), 12, 70001), (GameTable(dsl_tt), 4, 50000), (GameTable(dsl_ww), 8, 33333)]
    total = sum(r for _, _, r in segs)
    with RoomBatch(segs, seed=21, first_room=5, restart=True) as b:
        b.step(37)
        monkeypatch.setenv("GE_IO_THREADS", "1")
        one = b.read_rooms()
        monkeypatch.setenv("GE_IO_THREADS", "7")
        many = b.read_rooms()
        assert len(one) == total and one.tobytes() == many.tobytes()
        out = np.frombuffer(bytearray(b"\xAB" * (60000 * one.itemsize)), dtype=one.dtype)
        part = b.read_rooms(65000, 60000, out=out)           # spans all three segments
        assert part is out and part.tobytes() == one[65000:125000].tobytes()
        assert b.read_rooms(total - 1, 1).tobytes() == one[-1:].tobytes()
        # write -> read is the identity, whatever the thread count, also for a sub-range
        b.step(5)
        later = b.read_rooms()
        assert later.tobytes() != one.tobytes()
        b.write_rooms(0, one)
        assert b.read_rooms().tobytes() == one.tobytes()
        b.write_rooms(69990, later[69990:120011].copy())
        mixed = one.copy(); mixed[69990:120011] = later[69990:120011]
        assert b.read_rooms().tobytes() == mixed.tobytes()
        # one bad view in the last segment: GE_ERR_ARG, and no room of any segment was touched
        bad = later.copy()
        bad["players"][total - 7, 0, 0] = 9                  # werewolf role class out of range
        with pytest.raises(GeError) as e:
            b.write_rooms(0, bad)
        assert e.value.status == -1 and f"room {total - 7} " in str(e.value)       # ge_last_rejected_room names the view
        bad = later.copy()
        bad["n_players"][80000] = 5                          # a view that does not fit its segment
        bad["n_players"][90000] = 5                          # ... and a later one: the FIRST is reported
        with pytest.raises(GeError) as e:
            b.write_rooms(0, bad)
        assert e.value.status == -1 and "room 80000 " in str(e.value)
        with pytest.raises(GeError) as e:                    # a write that starts in the middle of the batch: the index is the batch's
            b.write_rooms(70000, bad[70000:100000].copy())
        assert "room 80000 " in str(e.value)
        # a Two-Truths view inside the Werewolf x 12 segment; phase ids no table row has (either of the two)
        for room, field, value in ((100, "pack", 2), (70001 + 17, "pack", 1), (total - 1, "phase_id", 1234),
                                   (3, "prev_phase_id", -5), (70001 + 49999, "phase_id", 16)):
            bad = later.copy()
            bad[field][room] = value
            with pytest.raises(GeError) as e:
                b.write_rooms(0, bad)
            assert e.value.status == -1 and f"room {room} " in str(e.value), (room, field)
        assert b.read_rooms().tobytes() == mixed.tobytes()
