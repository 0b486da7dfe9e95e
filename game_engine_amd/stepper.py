"""Host side of the batch stepper, mirroring the reference's interface for this path.

Reference surface kept (same names / argument meaning):
  load_dsl_by_gamename(gamename)            agent/tools/utils.py:557-581  (YAML -> dict)
  AgentState fields returned by agent_state current_phase_id, current_phase_name, player_states
                                            {"1": {<declared fields>}}  agent/game_agent_v2.py:97-117
  player ids "1".."N"                       agent/tools/utils.py:642-647
What is new: rooms are stepped in batches on the GPU (RoomBatch), one turn = one graph run.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib

ROOM_VIEW_DTYPE = np.dtype([("phase_id", "<i4"), ("prev_phase_id", "<i4"), ("end_turn", "<i4"), ("games", "<i4"),
                            ("phase0_done", "u1"), ("n_players", "u1"), ("pack", "u1"), ("pad", "u1"),
                            ("players", "u1", (16, 12)), ("det", "u1", (16,))])

EVENT_DTYPE = np.dtype([("turn", "<u4"), ("from_phase_id", "<i4"), ("to_phase_id", "<i4"), ("acted_now", "<u2"),
                        ("restarted", "u1"), ("pad", "u1"), ("choice", "u1", (16,))])

PACK_WEREWOLF, PACK_TWO_TRUTHS = 1, 2
WW_FIELDS = ["role", "team", "is_alive", "role_revealed", "can_vote", "has_secret_role",
             "night_action_eligible", "night_action_submitted", "selected_target_id"]
TT_FIELDS = ["is_speaker", "statements_submitted", "lie_index", "lie_revealed", "can_vote",
             "vote_choice", "has_voted", "total_score", "rounds_as_speaker"]
_TEAMS = ["", "villagers", "werewolves"]
# slots that are not ge_room_view columns (include/ge_step.h GE_WW_DET_MEMORY, GE_WW_WOLF_CHAT, GE_TT_STATEMENTS)
SLOT_DET_MEMORY, SLOT_WOLF_CHAT, SLOT_STATEMENTS = 9, 10, 9


GE_ERR_ARG = -1                           # include/ge_step.h ge_status


class GeError(RuntimeError):
    def __init__(self, status: int, what: str = ""):
        msg = _lib.load().ge_strerror(status).decode()
        super().__init__(f"{what}: {msg} ({status})" if what else f"{msg} ({status})")
        self.status = status


def library_path() -> str:
    return _lib.LIB_PATH


def _check(status: int, what: str = ""):
    if status != 0:
        raise GeError(status, what)


def load_dsl_by_gamename(gamename: str, games_dir: Optional[str] = None) -> dict:
    """Same contract as the reference's loader (utils.py:557-581): '<games_dir>/<gamename>.yaml'
    -> dict, {} when the name is empty or the file is missing.  Also accepts the JSON form of
    the same document ('<gamename>.json')."""
    if not gamename:
        return {}
    games_dir = games_dir or os.environ.get("GE_GAMES_DIR", "games")
    ypath = os.path.join(games_dir, f"{gamename}.yaml")
    jpath = os.path.join(games_dir, f"{gamename}.json")
    if os.path.exists(ypath):
        import yaml
        with open(ypath, encoding="utf-8") as f:
            return yaml.safe_load(f) or {}
    if os.path.exists(jpath):
        with open(jpath, encoding="utf-8") as f:
            return json.load(f)
    return {}


def initialize_player_states_from_dsl(dsl_content: dict, room_players: list) -> dict:
    """Same contract as the reference's helper (agent/tools/utils.py:584-653; TS twin
    src/app/api/games/initialize-players/route.ts:83-166): every room player gets a copy of
    declaration.player_states_template.player_states[<first id>] with its own `name`; ids "1".."N".
    Falls back to defaults derived from declaration.player_states when there is no template."""
    decl = (dsl_content or {}).get("declaration") or {}
    tmpl_all = (decl.get("player_states_template") or {}).get("player_states") or {}
    template = tmpl_all.get("1") or tmpl_all.get(1) or (tmpl_all[next(iter(tmpl_all))] if tmpl_all else {})
    if not template:
        defaults = {"string": "", "num": 0, "number": 0, "array": [], "list": [], "object": {}, "dict": {}}
        for name, spec in (decl.get("player_states") or {}).items():
            ftype, example = spec.get("type", "string"), spec.get("example")
            if ftype == "boolean":
                template[name] = example if example is not None else True
            else:
                template[name] = example or defaults.get(ftype)
    if not template:
        return {}
    return {str(i + 1): {**template, "name": p.get("name", f"Player {i + 1}")} for i, p in enumerate(room_players)}


class GameTable:
    """A game DSL compiled by ge_table_compile_json."""

    def __init__(self, dsl: dict, rounds: int = 1):
        if not isinstance(dsl, dict) or not dsl:
            raise GeError(-2, "empty DSL")
        self.dsl = dsl
        # declared fields the rule packs do not model: constants from the template (nobody writes them under the fixed policy)
        tmpl_all = ((dsl.get("declaration") or {}).get("player_states_template") or {}).get("player_states") or {}
        tmpl = tmpl_all.get("1") or tmpl_all.get(1) or (tmpl_all[next(iter(tmpl_all))] if tmpl_all else {})
        text = json.dumps(dsl, ensure_ascii=False).encode("utf-8")   # int phase keys become strings
        self.c = _lib.Table()
        err = C.create_string_buffer(512)
        st = _lib.load().ge_table_compile_json(text, len(text), rounds, C.byref(self.c), err, len(err))
        if st != 0:
            raise GeError(st, err.value.decode("utf-8", "replace"))
        # slot -> the DSL's own name for it ("" = the DSL does not declare the slot); include/ge_step.h GE_WW_* / GE_TT_*
        self.field_names: List[str] = [self.c.field_names[s].value.decode("utf-8", "replace") for s in range(_lib.GE_MAX_SLOTS)]
        modelled = set(n for n in self.field_names if n) | {"name"}
        self.extra_fields = {k: v for k, v in (tmpl or {}).items() if k not in modelled and isinstance(v, (bool, int, str))}

    @classmethod
    def from_gamename(cls, gamename: str, games_dir: Optional[str] = None, rounds: int = 1) -> "GameTable":
        return cls(load_dsl_by_gamename(gamename, games_dir), rounds)

    @property
    def pack(self) -> int:
        return self.c.pack

    @property
    def n_phases(self) -> int:
        return self.c.n_phases

    def rows(self) -> List[dict]:
        out = []
        for i in range(self.c.n_phases):
            r = self.c.rows[i]
            out.append({"phase_id": r.phase_id, "name": r.name.decode("utf-8", "replace"),
                        "completion": r.completion, "act": r.act, "effect": r.effect,
                        "terms": [(r.term_base[j], r.term_neg[j]) for j in range(r.n_terms)],
                        "generic": bool(r.generic),
                        # the condition in clause form: OR of AND-clauses of (kind, neg, bases bit set, num_field, lo, hi)
                        "clauses": [[(l.kind, l.neg, l.bases, l.num_field, l.lo, l.hi) for l in list(r.clause[c])[: r.clause_len[c]]]
                                    for c in range(r.n_clauses)],
                        "branches": [(r.br_res[j], r.br_target[j]) for j in range(r.n_branches)]})
        return out

    def phase_name(self, phase_id: int) -> str:
        for i in range(self.c.n_phases):
            if self.c.rows[i].phase_id == phase_id:
                return self.c.rows[i].name.decode("utf-8", "replace")
        return f"Phase {phase_id}"        # utils.py:30 fallback

    def role_name(self, cls_idx: int) -> str:
        return self.c.role_names[cls_idx].value.decode("utf-8", "replace")


Segment = Tuple                            # (table, n_players, n_rooms[, human_mask])


class RoomBatch:
    """A batch of independent rooms resident in HBM.  One `step()` = one turn of every room
    (= one LangGraph run per room in the reference)."""

    def __init__(self, segments: Sequence[Segment], seed: int = 0, first_room: int = 0,
                 device: int = 0, max_fuse: int = 0, restart: bool = False, trace: bool = False):
        lib = _lib.load()
        if not 1 <= len(segments) <= _lib.GE_MAX_SEGMENTS:
            raise GeError(-1, "segments")
        self.segments = list(segments)
        d = _lib.BatchDesc()
        d.seed, d.first_room, d.n_segments, d.device, d.max_fuse = seed, first_room, len(segments), device, max_fuse
        d.flags = (1 if restart else 0) | (2 if trace else 0)
        for k, seg in enumerate(segments):
            tb, n_players, n_rooms = seg[:3]
            d.seg[k].table = C.pointer(tb.c)
            d.seg[k].n_players, d.seg[k].n_rooms = n_players, n_rooms
            d.seg[k].human_mask = seg[3] if len(seg) > 3 else 0      # bit i: player i+1 is host-driven
        h = C.c_void_p()
        _check(lib.ge_batch_create(C.byref(d), C.byref(h)), "ge_batch_create")
        self._h = h
        self._lib = lib
        self._seed, self._first_room, self._max_fuse, self._restart, self._trace = seed, first_room, max_fuse, restart, trace
        self.n_rooms = sum(s[2] for s in segments)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ge_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- stepping
    def step(self, n_turns: int = 1, stream: int = 0):
        _check(self._lib.ge_batch_step(self._h, n_turns, C.c_void_p(stream or None)), "ge_batch_step")

    def reset(self):
        _check(self._lib.ge_batch_reset(self._h), "ge_batch_reset")

    def sync(self):
        _check(self._lib.ge_batch_sync(self._h), "ge_batch_sync")

    @property
    def turn(self) -> int:
        t = C.c_uint64()
        _check(self._lib.ge_batch_turn(self._h, C.byref(t)))
        return t.value

    def set_turn(self, turn: int):
        """Restore the turn counter of a checkpoint (the RNG and end_turn are keyed by it)."""
        _check(self._lib.ge_batch_set_turn(self._h, turn), "ge_batch_set_turn")

    def set_timing(self, on: bool):
        _check(self._lib.ge_batch_set_timing(self._h, int(on)))

    def kernel_time(self, reset: bool = True) -> Tuple[float, int]:
        ms, n = C.c_double(), C.c_uint64()
        _check(self._lib.ge_batch_kernel_time(self._h, int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- state access
    def read_rooms(self, first: int = 0, count: Optional[int] = None, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Canonical views of `count` rooms from `first` on.  `out`: an array of a previous call to fill again (a fresh
        228-byte-per-room array costs its page faults on top of the read)."""
        count = self.n_rooms - first if count is None else count
        if out is None:
            out = np.empty(count, dtype=ROOM_VIEW_DTYPE)        # the library writes every byte of every view
        assert out.dtype == ROOM_VIEW_DTYPE and out.flags.c_contiguous and len(out) == count
        _check(self._lib.ge_batch_read_rooms(self._h, first, count, out.ctypes.data, out.nbytes), "ge_batch_read_rooms")
        return out

    def write_rooms(self, first: int, views: np.ndarray):
        assert views.dtype == ROOM_VIEW_DTYPE and views.flags.c_contiguous
        st = self._lib.ge_batch_write_rooms(self._h, first, len(views), views.ctypes.data)
        if st == -1 and len(views):             # all-or-nothing: say which view did not fit its segment (pack / players / phase ids / role class)
            bad = int(self._lib.ge_last_rejected_room())
            if first <= bad < first + len(views):
                raise GeError(st, f"ge_batch_write_rooms: room {bad} does not fit its segment (nothing was written)")
        _check(st, "ge_batch_write_rooms")

    def inject_action(self, room: int, player_id: int, choice: int):
        """Log an action of a host-driven (human) player in the room's current phase."""
        _check(self._lib.ge_batch_inject_action(self._h, room, player_id, choice), "ge_batch_inject_action")

    def inject_actions(self, rooms, player_ids, choices) -> np.ndarray:
        """Log many host-driven players' actions at once (one kernel).  Returns the per-action status
        (0 = applied, negative ge_status = refused and nothing changed for that action)."""
        rooms = np.ascontiguousarray(rooms, dtype=np.uint64)
        player_ids = np.ascontiguousarray(player_ids, dtype=np.uint32)
        choices = np.ascontiguousarray(choices, dtype=np.uint32)
        if not (len(rooms) == len(player_ids) == len(choices)):
            raise GeError(-1, "inject_actions: arrays differ in length")
        status = np.zeros(len(rooms), dtype=np.int32)
        st = self._lib.ge_batch_inject_actions(self._h, len(rooms), rooms.ctypes.data, player_ids.ctypes.data,
                                               choices.ctypes.data, status.ctypes.data)
        # the return value is the first refused action's status - or a failure of the call itself (closed handle,
        # allocation, HIP error), which leaves the status array untouched: that one must not read as "all applied"
        if st != 0 and not status.any():
            _check(st, "ge_batch_inject_actions")
        return status

    def read_events(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        """[count, n_turns] events of the most recent step() call (batch created with trace=True)."""
        count = self.n_rooms - first if count is None else count
        n = C.c_uint32()
        st = self._lib.ge_batch_read_events(self._h, first, 0, C.byref(n), None, 0)
        _check(st, "ge_batch_read_events")
        out = np.zeros((count, max(n.value, 1)), dtype=EVENT_DTYPE)
        _check(self._lib.ge_batch_read_events(self._h, first, count, C.byref(n), out.ctypes.data, out.nbytes),
               "ge_batch_read_events")
        return out[:, : n.value]

    def summary(self) -> Dict[str, Any]:
        s = _lib.Summary()
        _check(self._lib.ge_batch_summary(self._h, C.byref(s)), "ge_batch_summary")
        return summary_to_dict(np.frombuffer(bytes(s), dtype="<u8"))

    def summary_words(self) -> np.ndarray:
        s = _lib.Summary()
        _check(self._lib.ge_batch_summary(self._h, C.byref(s)), "ge_batch_summary")
        return np.frombuffer(bytes(s), dtype="<u8").copy()

    # ---- checkpoint / resume (the reference delegates this to LangGraph thread persistence; here a batch
    # ---- is its room records + the turn counter, and the RNG is counter-based, so a resumed batch continues bit-exact)
    def save_checkpoint(self, path: str) -> None:
        """Self-contained, layout-independent checkpoint: every room as a ge_room_view, the turn counter, the batch
        description and the games' DSLs (numpy .npz; `load_checkpoint` rebuilds the batch from it alone)."""
        meta = {"abi": _lib.GE_ABI_VERSION, "seed": self._seed, "first_room": self._first_room, "turn": self.turn,
                "max_fuse": self._max_fuse, "restart": self._restart, "trace": self._trace,
                "segments": [{"dsl": seg[0].dsl, "rounds": int(seg[0].c.rounds), "n_players": int(seg[1]), "n_rooms": int(seg[2]),
                              "human_mask": int(seg[3]) if len(seg) > 3 else 0} for seg in self.segments]}
        with open(path, "wb") as f:
            np.savez_compressed(f, views=self.read_rooms(), meta=np.frombuffer(json.dumps(meta).encode("utf-8"), dtype=np.uint8))

    @classmethod
    def load_checkpoint(cls, path: str, device: int = 0) -> "RoomBatch":
        with np.load(path, allow_pickle=False) as z:
            meta = json.loads(bytes(z["meta"]).decode("utf-8"))
            views = np.ascontiguousarray(z["views"])
        if meta["abi"] != _lib.GE_ABI_VERSION:
            raise GeError(-1, f"checkpoint written by ABI {meta['abi']}, this library is ABI {_lib.GE_ABI_VERSION}")
        segs = [(GameTable(sg["dsl"], sg["rounds"]), sg["n_players"], sg["n_rooms"], sg["human_mask"]) for sg in meta["segments"]]
        b = cls(segs, seed=meta["seed"], first_room=meta["first_room"], device=device, max_fuse=meta["max_fuse"],
                restart=meta["restart"], trace=meta["trace"])
        b.write_rooms(0, views.astype(ROOM_VIEW_DTYPE, copy=False))
        b.set_turn(meta["turn"])
        return b

    def state(self, segment: int = 0) -> Tuple[int, int, int]:
        p, nbytes, bpr = C.c_void_p(), C.c_size_t(), C.c_uint32()
        _check(self._lib.ge_batch_state(self._h, segment, C.byref(p), C.byref(nbytes), C.byref(bpr)))
        return p.value, nbytes.value, bpr.value

    def bytes_per_room(self, segment: int = 0) -> int:
        return self.state(segment)[2]

    def _segment_of(self, room: int) -> Tuple[GameTable, int]:
        base = 0
        for seg in self.segments:
            tb, n, rooms = seg[:3]
            if room < base + rooms:
                return tb, n
            base += rooms
        raise IndexError(room)

    def agent_state(self, room: int) -> Dict[str, Any]:
        """AgentState-shaped dict of one room (agent/game_agent_v2.py:97-117): what the TS
        frontend's useCoAgent sync receives (src/lib/canvas/types.ts:338-360)."""
        tb, _ = self._segment_of(room)
        return view_to_agent_state(tb, self.read_rooms(room, 1)[0])


class RoomGroup:
    """One host process, several GPUs: the rooms of `segments` (the WHOLE job) sharded over `devices`, stepped
    concurrently, with one RCCL all-gather of the per-device summaries inside the native library (ge_group_*,
    include/ge_step.h).  Results equal those of one RoomBatch with the same arguments, for any number of devices.
    (The multi-process form, one rank per GPU over torch.distributed, is game_engine_amd.dist.)"""

    def __init__(self, segments: Sequence[Segment], devices: Sequence[int], seed: int = 0, first_room: int = 0,
                 max_fuse: int = 0, restart: bool = False, trace: bool = False):
        lib = _lib.load()
        if not 1 <= len(segments) <= _lib.GE_MAX_SEGMENTS:
            raise GeError(-1, "segments")
        self.segments = list(segments)
        d = self._desc = _job_desc(segments, seed, first_room, max_fuse, restart, trace)
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(lib.ge_group_create(C.byref(d), devs, len(devices), C.byref(h)), "ge_group_create")
        self._h, self._lib = h, lib
        self.n_devices = len(devices)
        self.n_rooms = sum(s[2] for s in segments)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ge_group_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def step(self, n_turns: int = 1):
        _check(self._lib.ge_group_step(self._h, n_turns), "ge_group_step")

    def sync(self):
        _check(self._lib.ge_group_sync(self._h), "ge_group_sync")

    def summary(self) -> Dict[str, Any]:
        s = _lib.Summary()
        _check(self._lib.ge_group_summary(self._h, C.byref(s)), "ge_group_summary")
        return summary_to_dict(np.frombuffer(bytes(s), dtype="<u8"))

    def shard_summaries(self) -> List[Dict[str, Any]]:
        """Each device's own ge_batch_summary (host-side cross-check of the collective: they add up to summary())."""
        out = []
        for i in range(self.n_devices):
            b, s = C.c_void_p(), _lib.Summary()
            _check(self._lib.ge_group_shard(self._h, i, C.byref(b)), "ge_group_shard")
            _check(self._lib.ge_batch_summary(b, C.byref(s)), "ge_batch_summary")
            out.append(summary_to_dict(np.frombuffer(bytes(s), dtype="<u8")))
        return out

    def read_rooms(self) -> np.ndarray:
        """All rooms in the order of one RoomBatch with the same segments (segment-major)."""
        shards = []
        for i in range(self.n_devices):
            b = C.c_void_p()
            _check(self._lib.ge_group_shard(self._h, i, C.byref(b)), "ge_group_shard")
            shards.append(b)
        return reassemble_rooms(self._lib, shards, self._desc)


def _job_desc(segments: Sequence[Segment], seed: int, first_room: int, max_fuse: int, restart: bool, trace: bool) -> "_lib.BatchDesc":
    d = _lib.BatchDesc()
    d.seed, d.first_room, d.n_segments, d.device, d.max_fuse = seed, first_room, len(segments), 0, max_fuse
    d.flags = (1 if restart else 0) | (2 if trace else 0)
    for k, seg in enumerate(segments):
        tb, n_players, n_rooms = seg[:3]
        d.seg[k].table = C.pointer(tb.c)
        d.seg[k].n_players, d.seg[k].n_rooms = n_players, n_rooms
        d.seg[k].human_mask = seg[3] if len(seg) > 3 else 0
    return d


def partition(desc: "_lib.BatchDesc", n_parts: int, part: int):
    """(shard desc, [global index of the part's first room of each segment]) - ge_group_partition, the arithmetic
    ge_group_create shards a job with: the part-th of n_parts contiguous parts of every segment.  No device is touched."""
    lib = _lib.load()
    shard = _lib.BatchDesc()
    first = (C.c_uint64 * _lib.GE_MAX_SEGMENTS)()
    _check(lib.ge_group_partition(C.byref(desc), n_parts, part, C.byref(shard), first), "ge_group_partition")
    return shard, [int(first[k]) for k in range(desc.n_segments)]


def reassemble_rooms(lib, shard_handles, desc: "_lib.BatchDesc") -> np.ndarray:
    """Every room of a sharded job in the order of ONE batch of `desc` (segment-major): shard i holds, segment by segment,
    the i-th part of each; read each part and put segment k's parts side by side.  Used by RoomGroup (devices of a node)
    and RoomShards (shards placed by the host)."""
    n = len(shard_handles)
    per_seg: List[List[np.ndarray]] = [[] for _ in range(desc.n_segments)]
    for i, b in enumerate(shard_handles):
        sd, _ = partition(desc, n, i)
        first = 0
        for k in range(desc.n_segments):
            cnt = int(sd.seg[k].n_rooms)
            out = np.empty(cnt, dtype=ROOM_VIEW_DTYPE)          # the library writes every byte of every view
            _check(lib.ge_batch_read_rooms(b, first, cnt, out.ctypes.data, out.nbytes), "ge_batch_read_rooms")
            per_seg[k].append(out)
            first += cnt
    return np.concatenate([x for seg in per_seg for x in seg])


class RoomShards:
    """The same sharding as RoomGroup, with the shards placed by the host: `devices[i]` is where part i of len(devices) lives -
    devices may repeat (several shards on one GPU) and no collective library is involved; summary() adds the shards' own
    summaries on the host (every field is a sum over rooms).  For hosts that schedule shards themselves, and the way the n > 1
    partition is exercised on a one-GPU box (tests/test_gpu_group.py)."""

    def __init__(self, segments: Sequence[Segment], devices: Sequence[int], seed: int = 0, first_room: int = 0,
                 max_fuse: int = 0, restart: bool = False, trace: bool = False):
        lib = _lib.load()
        if not 1 <= len(segments) <= _lib.GE_MAX_SEGMENTS:
            raise GeError(-1, "segments")
        self.segments, self._lib = list(segments), lib
        self._desc = _job_desc(segments, seed, first_room, max_fuse, restart, trace)
        self._h: List[C.c_void_p] = []
        self.firsts: List[List[int]] = []
        for i, dev in enumerate(devices):
            sd, first = partition(self._desc, len(devices), i)
            sd.device = dev
            h = C.c_void_p()
            arr = (C.c_uint64 * _lib.GE_MAX_SEGMENTS)(*(first + [0] * (_lib.GE_MAX_SEGMENTS - len(first))))
            st = lib.ge_batch_create_shard(C.byref(sd), arr, C.byref(h))
            if st != 0:
                self.close()
                _check(st, "ge_batch_create_shard")
            self._h.append(h)
            self.firsts.append(first)
        self.n_rooms = sum(s[2] for s in segments)

    def close(self):
        for h in getattr(self, "_h", []):
            self._lib.ge_batch_destroy(h)
        self._h = []

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def step(self, n_turns: int = 1):
        for h in self._h:                                  # asynchronous: shards on different devices step concurrently
            _check(self._lib.ge_batch_step(h, n_turns, None), "ge_batch_step")

    def shard_summaries(self) -> List[Dict[str, Any]]:
        out = []
        for h in self._h:
            s = _lib.Summary()
            _check(self._lib.ge_batch_summary(h, C.byref(s)), "ge_batch_summary")
            out.append(summary_to_dict(np.frombuffer(bytes(s), dtype="<u8")))
        return out

    def summary(self) -> Dict[str, Any]:
        return sum_summaries(self.shard_summaries())

    def read_rooms(self) -> np.ndarray:
        return reassemble_rooms(self._lib, self._h, self._desc)


def sum_summaries(parts: List[Dict[str, Any]]) -> Dict[str, Any]:
    """Whole-job summary from the shards' (what the all-gather + sum of ge_group_summary computes): sums mod 2^64; `turn` is common."""
    out: Dict[str, Any] = {}
    for k, v in parts[0].items():
        if isinstance(v, list):
            out[k] = [sum(p[k][j] for p in parts) & 0xFFFFFFFFFFFFFFFF for j in range(len(v))]
        else:
            out[k] = v if k == "turn" else sum(p[k] for p in parts) & 0xFFFFFFFFFFFFFFFF
    return out


def summary_to_dict(w: np.ndarray) -> Dict[str, Any]:
    w = [int(x) for x in w]
    return {"rooms": w[0], "finished": w[1], "village_wins": w[2], "wolf_wins": w[3], "alive_players": w[4],
            "sum_end_turn": w[5], "end_turn_hist": w[6:22], "score_hist": w[22:38], "checksum": w[38], "turn": w[39], "games_recycled": w[40]}


def project_view(view, table: Optional["GameTable"] = None) -> List[int]:
    """Canonical integer projection of one room: [phase, prev_phase, phase0_done, end_turn]
    + 11 ints per player (+ detective memory per player, werewolf) — the form the parity
    tests compare (tests/golden/*.json 'layout').  With `table`: only what the DSL declares (a slot it does not
    declare reads 0, as it does in a reference-run room, whose player_states have no such field)."""
    n = int(view["n_players"])
    out = [int(view["phase_id"]), int(view["prev_phase_id"]), int(view["phase0_done"]), int(view["end_turn"])]
    names = table.field_names if table is not None else None
    shown = [1 if (names is None or names[s]) else 0 for s in range(9)] + [1, 1]
    for i in range(n):
        out += [int(x) * k for x, k in zip(view["players"][i][:11], shown)]
    if int(view["pack"]) == PACK_WEREWOLF:
        det = 1 if (names is None or names[SLOT_DET_MEMORY]) else 0
        out += [int(x) * det for x in view["det"][:n]]
    return out


def slot_values(tb: GameTable, view, i: int) -> List[Any]:
    """Player i's state, one value per slot of the pack (GE_WW_* / GE_TT_* order), whether or not the DSL declares it."""
    f = [int(x) for x in view["players"][i]]
    if int(view["pack"]) == PACK_WEREWOLF:
        n = int(view["n_players"])
        mem = {str(k + 1): _TEAMS[int(d)] for k, d in enumerate(view["det"][:n]) if d} if f[0] == 4 else {}
        return [tb.role_name(f[0]), _TEAMS[f[1]], bool(f[2]), bool(f[3]), bool(f[4]), bool(f[5]), bool(f[6]), bool(f[7]), f[8],
                mem, f[1] == 2]
    return [bool(f[0]), bool(f[1]), f[2], bool(f[3]), bool(f[4]), f[5], bool(f[6]), f[7], f[8]]


def view_to_agent_state(tb: GameTable, view) -> Dict[str, Any]:
    """player_states hold exactly the fields the DSL declares, under the DSL's own names (GameTable.field_names)."""
    n = int(view["n_players"])
    ps: Dict[str, Dict[str, Any]] = {}
    for i in range(n):
        vals = slot_values(tb, view, i)
        ps[str(i + 1)] = {tb.field_names[s]: v for s, v in enumerate(vals) if tb.field_names[s]}
        ps[str(i + 1)].update(tb.extra_fields)             # declared fields no rule writes: the template's values
    pid = int(view["phase_id"])
    return {"current_phase_id": pid, "current_phase_name": tb.phase_name(pid), "player_states": ps,
            "previous_phase_id": int(view["prev_phase_id"]), "end_turn": int(view["end_turn"])}
