"""Helpers shared by the GPU parity tests: oracle rooms -> the product's room-view layout."""
import numpy as np

from game_engine_amd.stepper import ROOM_VIEW_DTYPE


def oracle_rooms_as_views(orc, rooms: np.ndarray) -> np.ndarray:
    """oracle.oracle.ROOM_DTYPE array -> ROOM_VIEW_DTYPE array (vectorised), for whole-batch memcmp."""
    ids = np.array(orc.ids, dtype=np.int32)
    v = np.zeros(len(rooms), dtype=ROOM_VIEW_DTYPE)
    v["phase_id"] = ids[rooms["phase"]]
    v["prev_phase_id"] = ids[rooms["prev"]]
    v["end_turn"] = rooms["end_turn"]
    v["games"] = rooms["games"]
    v["phase0_done"] = rooms["phase0_done"]
    v["n_players"] = rooms["n"]
    v["pack"] = orc.table.pack
    v["players"] = rooms["p"]
    v["players"][:, :, 11] = 0
    v["det"] = rooms["det"]
    return v


def oracle_batch(orc, n_rooms, seed, first_room, turns, threads=0, restart=False):
    rooms = orc.init_rooms(n_rooms)
    orc.run(rooms, seed, first_room, 0, turns, threads=threads, restart=restart)
    return oracle_rooms_as_views(orc, rooms)


def assert_views_equal(got: np.ndarray, want: np.ndarray, what=""):
    if got.tobytes() == want.tobytes():
        return
    for name in ROOM_VIEW_DTYPE.names:
        bad = np.nonzero((got[name] != want[name]).reshape(len(got), -1).any(axis=1))[0]
        if len(bad):
            i = int(bad[0])
            raise AssertionError(f"{what}: field {name!r} differs in {len(bad)} rooms; first room {i}: "
                                 f"got {got[name][i].tolist()} want {want[name][i].tolist()}")
    raise AssertionError(what + ": padding differs")


def oracle_events(orc, rooms: np.ndarray, turn: int) -> np.ndarray:
    """The oracle's record of the turn it just ran, in the product's ge_turn_event layout."""
    from game_engine_amd.stepper import EVENT_DTYPE
    ids = np.array(orc.ids, dtype=np.int32)
    e = np.zeros(len(rooms), dtype=EVENT_DTYPE)
    e["turn"] = turn
    e["from_phase_id"] = ids[rooms["ev_from"]]
    e["to_phase_id"] = ids[rooms["ev_to"]]
    e["acted_now"] = rooms["ev_newly"]
    e["restarted"] = rooms["ev_restarted"]
    e["choice"] = rooms["ev_choice"]
    return e


def views_as_oracle_rooms(orc, views: np.ndarray) -> np.ndarray:
    """ROOM_VIEW_DTYPE array -> oracle ROOM_DTYPE array (inverse of oracle_rooms_as_views)."""
    from oracle.oracle import ROOM_DTYPE
    idx_of = {pid: i for i, pid in enumerate(orc.ids)}
    r = np.zeros(len(views), dtype=ROOM_DTYPE)
    r["phase"] = [idx_of[int(x)] for x in views["phase_id"]]
    r["prev"] = [idx_of[int(x)] for x in views["prev_phase_id"]]
    r["phase0_done"] = views["phase0_done"]
    r["n"] = views["n_players"]
    r["end_turn"] = views["end_turn"]
    r["games"] = views["games"]
    r["p"] = views["players"]
    r["det"] = views["det"]
    return r
