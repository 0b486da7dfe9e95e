"""ORACLE (test infrastructure) — ctypes front of oracle/ge_oracle.c.

`Oracle(dsl_dict, n_players)` compiles the DSL with oracle/dsl_table.py, feeds the
table to the C restatement and returns canonical projections
([phase_id, prev_phase_id, phase0_done, end_turn] + 11 ints per player
 (+ detective memory per player for the werewolf pack)) — the same list the
reference harness (oracle/refharness/walker.py:project_state) produces.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List

import numpy as np

from . import dsl_table as T

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libge_oracle.so")


class _Literal(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("neg", C.c_uint8), ("num_field", C.c_uint8), ("pad", C.c_uint8),
                ("bases", C.c_uint16), ("lo", C.c_uint8), ("hi", C.c_uint8)]


class _Phase(C.Structure):
    _fields_ = [("completion", C.c_uint8), ("act", C.c_uint8), ("effect", C.c_uint8),
                ("n_terms", C.c_uint8), ("n_branches", C.c_uint8), ("pad", C.c_uint8 * 3),
                ("term_base", C.c_uint8 * 4), ("term_neg", C.c_uint8 * 4),
                ("br_res", C.c_uint8 * 4), ("br_target", C.c_uint8 * 4),
                ("phase_id", C.c_int32),
                ("n_clauses", C.c_uint8), ("clause_len", C.c_uint8 * 4), ("pad2", C.c_uint8 * 3),
                ("clause", (_Literal * 4) * 4)]


class _Table(C.Structure):
    _fields_ = [("pack", C.c_int32), ("n_phases", C.c_int32), ("rounds", C.c_int32), ("pad", C.c_int32),
                ("init_fields", C.c_uint8 * 12), ("pad2", C.c_uint8 * 4), ("ph", _Phase * 32)]


ROOM_DTYPE = np.dtype([("phase", "u1"), ("prev", "u1"), ("phase0_done", "u1"), ("n", "u1"),
                       ("end_turn", "<i4"), ("games", "<i4"), ("p", "u1", (16, 12)), ("det", "u1", (16,)),
                       ("ev_from", "u1"), ("ev_to", "u1"), ("ev_restarted", "u1"), ("ev_pad", "u1"),
                       ("ev_newly", "<u2"), ("ev_pad2", "<u2"), ("ev_choice", "u1", (16,))])


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ge_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libge_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_room_init.argtypes = [C.POINTER(_Table), C.c_int, C.c_void_p]
        _lib.orc_run.argtypes = [C.POINTER(_Table), C.c_uint64, C.c_uint64, C.c_uint64,
                                 C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_uint32]
        _lib.orc_inject_action.argtypes = [C.POINTER(_Table), C.c_void_p, C.c_int, C.c_int]
        assert _lib.orc_sizeof_room() == ROOM_DTYPE.itemsize
        assert _lib.orc_sizeof_table() == C.sizeof(_Table)
    return _lib


def _init_fields(table: T.Table) -> List[int]:
    tmpl = table.template

    class _Bound(dict):          # the template read through the schema binding (canonical slot -> declared name)
        def get(self, slot, default=None):
            name = table.declared(slot)
            return tmpl.get(name, default) if name else default
    t = _Bound()
    if table.pack == T.PACK_WEREWOLF:
        team = {"": 0, "villagers": 1, "werewolves": 2}[t.get("team", "")]
        role = table.role_names.index(t.get("role", "")) if t.get("role", "") in table.role_names else 0
        f = [role, team, int(bool(t.get("is_alive", True))), int(bool(t.get("role_revealed"))),
             int(bool(t.get("can_vote"))), int(bool(t.get("has_secret_role"))),
             int(bool(t.get("night_action_eligible"))), int(bool(t.get("night_action_submitted"))),
             int(t.get("selected_target_id") or 0), 0, 0]
    else:
        f = [int(bool(t.get("is_speaker"))), int(bool(t.get("statements_submitted"))),
             int(t.get("lie_index") or 0), int(bool(t.get("lie_revealed"))), int(bool(t.get("can_vote"))),
             int(t.get("vote_choice") or 0), int(bool(t.get("has_voted"))), int(t.get("total_score") or 0),
             int(t.get("rounds_as_speaker") or 0), 0, 0]
    return f + [0]


_CORES = None


def usable_cores() -> int:
    """Threads this process may run at once: the affinity mask, capped by the cgroup CPU quota."""
    global _CORES
    if _CORES is None:
        n = len(os.sched_getaffinity(0))
        for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            try:
                with open(path) as f:
                    parts = f.read().split()
                if path.endswith("cpu.max"):
                    if parts[0] != "max":
                        n = min(n, max(1, -(-int(parts[0]) // int(parts[1]))))
                else:
                    q = int(parts[0])
                    if q > 0:
                        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                            n = min(n, max(1, -(-q // int(g.read().split()[0]))))
                break
            except (OSError, ValueError, IndexError):
                continue
        _CORES = n
    return _CORES


class Oracle:
    def __init__(self, dsl: dict, n_players: int, rounds: int = 1):
        self.table = T.compile_dsl(dsl, rounds=rounds)
        self.n = n_players
        lo = 4 if self.table.pack == T.PACK_WEREWOLF else 3        # declaration.min_players
        if not lo <= n_players <= 12:
            raise ValueError("n_players out of range")
        ct = _Table()
        ct.pack, ct.n_phases, ct.rounds = self.table.pack, len(self.table.phases), rounds
        for i, v in enumerate(_init_fields(self.table)):
            ct.init_fields[i] = v
        for p in self.table.phases:
            cp = ct.ph[p.idx]
            cp.completion, cp.act, cp.effect = p.completion, p.act, p.effect
            cp.n_terms, cp.n_branches, cp.phase_id = len(p.terms), len(p.branches), p.id
            for k, t in enumerate(p.terms):
                cp.term_base[k], cp.term_neg[k] = t.base, int(t.negate)
            cp.n_clauses = len(p.clauses)
            for ci, clause in enumerate(p.clauses):
                cp.clause_len[ci] = len(clause)
                for li, l in enumerate(clause):
                    cl = cp.clause[ci][li]
                    cl.kind, cl.neg = (1 if l.kind == "base" else 2), int(l.negate)
                    cl.num_field, cl.lo, cl.hi = l.num, l.lo, l.hi
                    cl.bases = sum(1 << b for b in l.bases)
            for k, b in enumerate(p.branches):
                cp.br_res[k], cp.br_target[k] = b.resolver, b.target_idx
        self.ct = ct
        self.ids = [p.id for p in self.table.phases]

    def init_rooms(self, n_rooms: int) -> np.ndarray:
        rooms = np.zeros(n_rooms, dtype=ROOM_DTYPE)
        one = np.zeros(1, dtype=ROOM_DTYPE)
        lib().orc_room_init(C.byref(self.ct), self.n, one.ctypes.data)
        rooms[:] = one[0]
        return rooms

    def run(self, rooms: np.ndarray, seed: int, first_room: int, first_turn: int, n_turns: int,
            threads: int = 1, restart: bool = False, human_mask: int = 0) -> None:
        """threads = 0: as many as this process may really run at once (affinity mask capped by the cgroup CPU quota - a
        GPU box grants 16 cores of a 256-thread host, and an OpenMP team of 256 costs 0.1 s per call), and no more than
        the work is worth (one per 4 096 room-turns)."""
        assert rooms.dtype == ROOM_DTYPE and rooms.flags.c_contiguous
        if threads == 0:
            threads = max(1, min(usable_cores(), len(rooms) * max(n_turns, 1) // 4096))
        lib().orc_run(C.byref(self.ct), seed, first_room, len(rooms), first_turn, n_turns,
                      rooms.ctypes.data, threads, int(restart), human_mask)

    def inject(self, rooms: np.ndarray, index: int, player_id: int, choice: int) -> bool:
        """Log a host-driven player's action in room `index`; False if not allowed."""
        ptr = rooms.ctypes.data + index * ROOM_DTYPE.itemsize
        return lib().orc_inject_action(C.byref(self.ct), ptr, player_id, choice) == 0

    def project(self, room, declared_only: bool = False) -> List[int]:
        """The room as integers.  Default: the whole record (what the product's read_rooms is compared with);
        `declared_only`: the reference-run form of the goldens - a slot the DSL does not declare reads 0."""
        out = [self.ids[int(room["phase"])], self.ids[int(room["prev"])], int(room["phase0_done"]),
               int(room["end_turn"])]
        # a slot the DSL does not declare is not part of player_states: 0 in the projection, as in walker.project_state
        slots = T.WW_SLOTS if self.table.pack == T.PACK_WEREWOLF else T.TT_SLOTS
        shown = [1 if (self.table.declared(s[0]) or not declared_only) else 0 for s in slots[:9]] + [1, 1]
        for i in range(self.n):
            out += [int(x) * k for x, k in zip(room["p"][i][:11], shown)]
        if self.table.pack == T.PACK_WEREWOLF:
            det = 1 if (self.table.declared("investigated_alignments") or not declared_only) else 0
            out += [int(x) * det for x in room["det"][: self.n]]
        return out

    def trajectory(self, seed: int, room_index: int, n_turns: int, restart: bool = False,
                   first_turn: int = 0, human_mask: int = 0, human=None) -> List[List[int]]:
        """`human(turn, projection) -> (player_id, choice) | None` is asked before every turn."""
        rooms = self.init_rooms(1)
        out = []
        for t in range(first_turn, first_turn + n_turns):
            if human is not None:
                act = human(t, self.project(rooms[0]))
                if act:
                    assert self.inject(rooms, 0, act[0], act[1]), (t, act)
            self.run(rooms, seed, room_index, t, 1, restart=restart, human_mask=human_mask)
            out.append(self.project(rooms[0], declared_only=True))
        return out
