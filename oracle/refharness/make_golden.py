"""Generates tests/golden/*.json by running the REFERENCE's node coroutines
(/root/reference/agent/game_agent_v2.py, and v3 as a cross-check) under FixedPolicy.

Build container only (needs /root/reference).  Run:  python -m oracle.refharness.make_golden
Outputs (committed; data only — inputs and expected integer projections):
  tests/golden/dsl/<game>.json          yaml.safe_load of /root/reference/games/<game>.yaml
  tests/golden/traj_<game>_n<N>.json    per (seed, room): projection after every turn
"""
from __future__ import annotations

import json
import os
import sys

import yaml

from .walker import REFERENCE_ROOT, RoomSession

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
SEEDS = [0, 1, 0xC0FFEE]
CASES = [                                # (game, n_players, rooms, turns, rounds)
    ("werewolf-(mafia)", 8, [0, 1, 2, 65535], 72, 1),
    ("werewolf-(mafia)", 12, [0, 1, 16777215], 96, 1),
    ("werewolf-(mafia)", 4, [0, 5], 48, 1),
    ("werewolf-(mafia)", 5, [0, 3], 48, 1),
    ("werewolf-(mafia)", 7, [2], 64, 1),
    ("werewolf-(mafia)", 6, [8], 64, 1),
    ("werewolf-(mafia)", 9, [11], 96, 1),
    ("werewolf-(mafia)", 10, [3], 110, 1),
    ("werewolf-(mafia)", 11, [6], 110, 1),
    ("two-truths-and-a-lie", 4, [0, 1, 2, 1048575], 64, 1),
    ("two-truths-and-a-lie", 3, [0, 9], 48, 1),
    ("two-truths-and-a-lie", 6, [4], 160, 2),
    ("two-truths-and-a-lie", 12, [1], 200, 1),
    ("two-truths-and-a-lie", 5, [2], 260, 3),
    ("two-truths-and-a-lie", 8, [7], 150, 1),
    # the reference's earlier Werewolf draft (game_draft/): same rules, its own field names, no per-player target field
    ("draft-werewolf-(mafia)", 8, [0, 3, 65535], 90, 1),
    ("draft-werewolf-(mafia)", 5, [1], 60, 1),
    ("draft-werewolf-(mafia)", 12, [2], 120, 1),
    ("draft-werewolf-(mafia)", 4, [0], 40, 1),
    ("draft-werewolf-(mafia)", 9, [4], 110, 1),
]


def dsl_path(game: str) -> str:
    return os.path.join(REFERENCE_ROOT, "game_draft", f"{game[6:]}.yaml") if game.startswith("draft-") else \
        os.path.join(REFERENCE_ROOT, "games", f"{game}.yaml")


def file_tag(game: str) -> str:
    return game.split("-(")[0].replace("-", "_")


def main(only=None):
    os.makedirs(os.path.join(GOLD, "dsl"), exist_ok=True)
    for game in sorted({c[0] for c in CASES}):
        with open(dsl_path(game), encoding="utf-8") as f:
            dsl = yaml.safe_load(f)
        with open(os.path.join(GOLD, "dsl", f"{game}.json"), "w", encoding="utf-8") as f:
            json.dump(dsl, f, ensure_ascii=False, indent=1)
    for game, n, rooms, turns, rounds in CASES:
        cases = []
        for seed in SEEDS:
            for room in rooms:
                a = RoomSession(game, n, seed, room, "v2", rounds)
                b = RoomSession(game, n, seed, room, "v3", rounds)
                traj = []
                for t in range(turns):
                    a.step()
                    b.step()
                    pa = a.project()
                    assert pa == b.project(), (game, n, seed, room, t, "v2 != v3")
                    traj.append(pa)
                assert traj[-1][3] >= 0, (game, n, seed, room, "did not finish; raise turns")
                cases.append({"seed": seed, "room": room, "turns": traj})
                print(game, n, hex(seed), room, "end_turn", traj[-1][3], file=sys.stderr)
        name = f"traj_{file_tag(game)}_n{n}.json"
        if only and name not in only:
            continue
        with open(os.path.join(GOLD, name), "w") as f:
            json.dump({"game": game, "n_players": n, "rounds": rounds,
                       "source": "reference game_agent_v2 + v3 nodes under oracle/refharness FixedPolicy",
                       "layout": "[phase, prev_phase, phase0_done, end_turn] + 11/player (+ det/player for werewolf)",
                       "cases": cases}, f, separators=(",", ":"))


def restart_cases():
    """Steady-state mode (GE_FLAG_RESTART): when a room is finished, the NEXT turn belongs to a
    brand-new room on the same slot whose clock starts at that turn.  Reference-side that is a
    fresh RoomSession(turn0=...) — a new LangGraph thread — chained after the finished one."""
    out = []
    for game, n, turns in (("werewolf-(mafia)", 8, 260), ("two-truths-and-a-lie", 4, 200),
                           ("werewolf-(mafia)", 12, 300), ("draft-werewolf-(mafia)", 8, 260)):
        cases = []
        for seed in SEEDS:
            room = 77
            sess = {v: RoomSession(game, n, seed, room, v) for v in ("v2", "v3")}
            traj, games = [], 0
            for t in range(turns):
                if sess["v2"].end_turn >= 0:
                    games += 1
                    sess = {v: RoomSession(game, n, seed, room, v, turn0=t, game_index=games) for v in ("v2", "v3")}
                for s_ in sess.values():
                    s_.step()
                pa = sess["v2"].project()
                assert pa == sess["v3"].project()
                traj.append(pa)
            assert games >= 3
            cases.append({"seed": seed, "room": room, "turns": traj, "games": games})
            print("restart", game, n, hex(seed), "games", games, file=sys.stderr)
        name = f"restart_{file_tag(game)}_n{n}.json"
        with open(os.path.join(GOLD, name), "w") as f:
            json.dump({"game": game, "n_players": n, "rounds": 1, "restart": True,
                       "source": "chained reference sessions (v2 + v3) under FixedPolicy, clock continuing",
                       "cases": cases}, f, separators=(",", ":"))


def human_cases():
    """Player 1 is host-driven (the reference's human): the bot policy skips it and a scripted
    person (oracle/human_script.py) acts for it."""
    from ..human_script import HUMAN_MASK, person
    for game, n, turns in (("werewolf-(mafia)", 8, 110), ("two-truths-and-a-lie", 4, 100), ("werewolf-(mafia)", 12, 140),
                           ("draft-werewolf-(mafia)", 8, 110)):
        cases = []
        for seed in SEEDS:
            room = 31
            sess = {v: RoomSession(game, n, seed, room, v, human_mask=HUMAN_MASK, human_script=person) for v in ("v2", "v3")}
            traj, sent = [], 0
            for t in range(turns):
                for s_ in sess.values():
                    s_.step()
                assert sess["v2"].last_message == sess["v3"].last_message
                sent += sess["v2"].last_message != "Continue"
                pa = sess["v2"].project()
                assert pa == sess["v3"].project()
                traj.append(pa)
            assert traj[-1][3] >= 0, (game, n, seed, "did not finish")
            # every message of the person went through the reference's own process_human_action_if_needed
            filed = sess["v2"].state["playerActions"].get("1", {}).get("actions", {})
            assert len(filed) == sent and all(a["phase"] == sess["v2"].table.by_id(0).name for a in filed.values())
            cases.append({"seed": seed, "room": room, "turns": traj})
            print("human", game, n, hex(seed), "end", traj[-1][3], "messages", sent, file=sys.stderr)
        name = f"human_{file_tag(game)}_n{n}.json"
        with open(os.path.join(GOLD, name), "w") as f:
            json.dump({"game": game, "n_players": n, "rounds": 1, "human_mask": HUMAN_MASK,
                       "source": "reference v2 + v3 nodes under FixedPolicy with player 1 host-driven: the scripted person "
                                 "(oracle/human_script.py) sends the frontend's message strings, the reference's own "
                                 "process_human_action_if_needed logs them, the policy reads the choice off that entry",
                       "cases": cases}, f, separators=(",", ":"))


def variant_cases():
    """The reference's nodes run on VARIANTS of the shipped DSLs that use the rest of the condition grammar
    (oracle/dsl_variants.py: in [...], !=, <, <=, >, >=, or, wait_for kinds) -> traj_variant_<name>_n<N>.json."""
    from .. import dsl_variants
    for name, n, rooms, turns in (("ww_generic", 8, [0, 5], 90), ("ww_generic", 11, [2], 130),
                                  ("tt_generic", 4, [0, 3], 110), ("tt_generic", 7, [1], 200),
                                  ("ww_extra_fields", 8, [0, 4], 90), ("ww_minimal_schema", 8, [0, 7], 90),
                                  ("ww_minimal_schema", 12, [3], 130)):
        game, builder, rounds = dsl_variants.VARIANTS[name]
        cases = []
        for seed in SEEDS:
            for room in rooms:
                a = RoomSession(game, n, seed, room, "v2", rounds, dsl_variant=builder)
                b = RoomSession(game, n, seed, room, "v3", rounds, dsl_variant=builder)
                traj = []
                for t in range(turns):
                    a.step()
                    b.step()
                    pa = a.project()
                    assert pa == b.project(), (name, n, seed, room, t, "v2 != v3")
                    traj.append(pa)
                assert traj[-1][3] >= 0, (name, n, seed, room, "did not finish; raise turns")
                cases.append({"seed": seed, "room": room, "turns": traj})
                print("variant", name, n, hex(seed), room, "end_turn", traj[-1][3], file=sys.stderr)
        with open(os.path.join(GOLD, f"traj_variant_{name}_n{n}.json"), "w") as f:
            json.dump({"game": game, "variant": name, "n_players": n, "rounds": rounds,
                       "source": "reference game_agent_v2 + v3 nodes under FixedPolicy on oracle/dsl_variants.py:" + name,
                       "layout": "[phase, prev_phase, phase0_done, end_turn] + 11/player (+ det/player for werewolf)",
                       "cases": cases}, f, separators=(",", ":"))


def _strip_ts(x):
    """Drop wall-clock fields (timestamps) - everything else of the log-shaped AgentState parts stays."""
    if isinstance(x, dict):
        return {k: _strip_ts(v) for k, v in x.items() if k != "timestamp"}
    if isinstance(x, list):
        return [_strip_ts(v) for v in x]
    return x


def string_cases():
    """The STRING layer of AgentState (v2:97-117) as the reference run leaves it, turn by turn, for a few
    rooms: playerActions (bt:285-344), game_notes (bt:163-202), phase_history (v2:1207-1215) and the full
    player_states dicts (names, role / team strings, statements) - timestamps stripped.  What the hosts'
    RoomService must reproduce exactly (tests/test_strings_golden.py)."""
    for game, n, picks in (("werewolf-(mafia)", 8, [(0xC0FFEE, 0), (1, 3)]),
                           ("werewolf-(mafia)", 12, [(0xC0FFEE, 1)]),
                           ("two-truths-and-a-lie", 4, [(0xC0FFEE, 0), (0, 2)]),
                           ("two-truths-and-a-lie", 6, [(1, 4)]),
                           ("draft-werewolf-(mafia)", 8, [(0xC0FFEE, 0), (0, 5)])):
        cases = []
        for seed, room in picks:
            s_ = RoomSession(game, n, seed, room, "v2")
            turns, n_notes, n_hist = [], 0, 0
            seen_actions = {}
            t = 0
            while s_.end_turn < 0 or t < s_.end_turn + 3:
                s_.step()
                st = s_.state
                acts = []
                for pid in sorted(st["playerActions"], key=int):
                    rec = st["playerActions"][pid]
                    for aid in sorted(rec["actions"], key=int):
                        if (pid, aid) not in seen_actions:
                            seen_actions[(pid, aid)] = True
                            a = rec["actions"][aid]
                            acts.append({"player_id": pid, "name": rec["name"], "id": a["id"], "action": a["action"], "phase": a["phase"]})
                turns.append({"current_phase_id": st["current_phase_id"], "current_phase_name": st.get("current_phase_name"),
                              "actions_added": acts, "notes_added": list(st["game_notes"][n_notes:]),
                              "history_added": _strip_ts(st["phase_history"][n_hist:]),
                              "player_states": _strip_ts(st["player_states"])})
                n_notes, n_hist = len(st["game_notes"]), len(st["phase_history"])
                t += 1
                assert t < 400
            cases.append({"seed": seed, "room": room, "turns": turns,
                          "final": {"playerActions": _strip_ts(s_.state["playerActions"]), "game_notes": s_.state["game_notes"],
                                    "phase_history": _strip_ts(s_.state["phase_history"])}})
            print("strings", game, n, hex(seed), room, "turns", len(turns), file=sys.stderr)
        name = f"strings_{file_tag(game)}_n{n}.json"
        with open(os.path.join(GOLD, name), "w", encoding="utf-8") as f:
            json.dump({"game": game, "n_players": n, "rounds": 1,
                       "source": "reference game_agent_v2 nodes + backend_tools plumbing under FixedPolicy; timestamps stripped",
                       "cases": cases}, f, ensure_ascii=False, separators=(",", ":"))


def _turn_record(st, seen_actions, n_notes, n_hist):
    acts = []
    for pid in sorted(st["playerActions"], key=int):
        rec = st["playerActions"][pid]
        for aid in sorted(rec["actions"], key=int):
            if (pid, aid) not in seen_actions:
                seen_actions[(pid, aid)] = True
                a = rec["actions"][aid]
                acts.append({"player_id": pid, "name": rec["name"], "id": a["id"], "action": a["action"], "phase": a["phase"]})
    return {"current_phase_id": st["current_phase_id"], "current_phase_name": st.get("current_phase_name"),
            "actions_added": acts, "notes_added": list(st["game_notes"][n_notes:]),
            "history_added": _strip_ts(st["phase_history"][n_hist:]),
            "player_states": _strip_ts(st["player_states"])}


def string_human_cases():
    """The string layer of rooms with a PERSON in them, message by message: what the browser sent (vote / input / button /
    chat / control strings of src/app/page.tsx, oracle/human_script.py::talkative_person), whether a turn was played, and the
    reference's AgentState after it - the person's log entries as the reference's own process_human_action_if_needed files
    them (Player 1, phase 0's name, 200 characters).  What RoomService.handle_message must reproduce."""
    from ..human_script import talkative_person
    for game, n, picks in (("werewolf-(mafia)", 8, [(0xC0FFEE, 2, (1,)), (1, 6, (1,)), (0, 9, (1, 3))]),
                           ("two-truths-and-a-lie", 4, [(0xC0FFEE, 0, (1,)), (1, 5, (1, 3))]),
                           ("draft-werewolf-(mafia)", 8, [(0, 4, (1,))])):
        cases = []
        for seed, room, seats in picks:
            mask = sum(1 << (k - 1) for k in seats)
            names = [("Alice" if i == 0 else "Carol" if i == 2 else f"Bot {i + 1}") for i in range(n)]
            s_ = RoomSession(game, n, seed, room, "v2", human_mask=mask, human_script=talkative_person(seats), names=names)
            b_ = RoomSession(game, n, seed, room, "v3", human_mask=mask, human_script=talkative_person(seats), names=names)
            msgs, n_notes, n_hist, seen = [], 0, 0, {}
            while s_.end_turn < 0 or s_.turn < s_.end_turn + 3:
                played = s_.step()
                assert b_.step() == played and b_.last_message == s_.last_message and b_.project() == s_.project()
                rec = _turn_record(s_.state, seen, n_notes, n_hist)
                rec.update(message=s_.last_message, played=played)
                if not played:
                    assert not rec["actions_added"] and not rec["notes_added"] and not rec["history_added"]
                msgs.append(rec)
                n_notes, n_hist = len(s_.state["game_notes"]), len(s_.state["phase_history"])
                assert len(msgs) < 600
            cases.append({"seed": seed, "room": room, "human_seats": list(seats), "names": names, "messages": msgs,
                          "final": {"playerActions": _strip_ts(s_.state["playerActions"]), "game_notes": s_.state["game_notes"],
                                    "phase_history": _strip_ts(s_.state["phase_history"])}})
            kinds = {}
            for m in msgs:
                k = m["message"].split(" ")[0]
                kinds[k] = kinds.get(k, 0) + 1
            print("strings_human", game, n, hex(seed), room, "messages", len(msgs), kinds, file=sys.stderr)
        name = f"strings_human_{file_tag(game)}_n{n}.json"
        with open(os.path.join(GOLD, name), "w", encoding="utf-8") as f:
            json.dump({"game": game, "n_players": n, "rounds": 1,
                       "source": "reference game_agent_v2 nodes (v3 asserted equal) under FixedPolicy with host-driven seats; the "
                                 "person's messages are the frontend's strings, logged by the reference's own "
                                 "process_human_action_if_needed; timestamps stripped",
                       "cases": cases}, f, ensure_ascii=False, separators=(",", ":"))


if __name__ == "__main__":
    main()
    restart_cases()
    human_cases()
    string_cases()
    string_human_cases()
    variant_cases()
