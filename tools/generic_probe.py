#!/usr/bin/env python3
"""What a DSL with generic target conditions (or / in [..] / numeric comparisons: the GENERIC kernel builds)
costs against the shipped game whose conditions are plain conjunctions.  The re-conditioned Werewolf below is the one
the reference-run goldens traj_variant_ww_generic_* were produced with.  python tools/generic_probe.py"""
import copy, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from game_engine_amd import GameTable, RoomBatch

with open(os.path.join(ROOT, "tests", "golden", "dsl", "werewolf-(mafia).json"), encoding="utf-8") as f:
    base = json.load(f)
gen = copy.deepcopy(base)
def cond(pid, c):
    gen["phases"][str(pid)]["completion_criteria"]["target_players"]["condition"] = c
for pid in (2, 10):
    cond(pid, "player.role in ['Werewolf'] and player.is_alive == true and player.team != 'villagers'")
for pid in (3, 11):
    cond(pid, "player.role in ['Doctor', 'Medic'] and player.is_alive != false")
for pid in (4, 12):
    cond(pid, "player.role == 'Detective' and player.is_alive == true and player.selected_target_id == 0 "
              "or player.role == 'Detective' and player.selected_target_id > 0")
for pid in (7, 15):
    cond(pid, "player.can_vote == true and player.is_alive == true and player.selected_target_id < 5")
with open(os.path.join(ROOT, "tests", "golden", "dsl", "two-truths-and-a-lie.json"), encoding="utf-8") as f:
    tt_base = json.load(f)
tt_gen = copy.deepcopy(tt_base)                        # the tt_generic variant of the goldens
def tcond(pid, c):
    tt_gen["phases"][str(pid)]["completion_criteria"]["target_players"]["condition"] = c
tcond(2, "player.is_speaker == true and player.statements_submitted != true")
tcond(3, "player.is_speaker in [true] and player.lie_index not in [1, 2, 3]")
tcond(5, "player.is_speaker == false and player.total_score <= 1 or player.is_speaker == false and player.rounds_as_speaker >= 1")
# the SHIPPED rules in generic form: every action phase's own condition with a numeric literal that always holds appended, so the
# rows run the clause form but the games are the same games - the pure cost of the evaluation, without a change of dynamics
same, tt_same = copy.deepcopy(base), copy.deepcopy(tt_base)
for d, extra in ((same, " and player.selected_target_id >= 0"), (tt_same, " and player.total_score >= 0")):
    for ph in d["phases"].values():
        cc = ph.get("completion_criteria") or {}
        if cc.get("type") == "player_action":
            cc["target_players"]["condition"] += extra
# the same games again with FOUR literal slots (2 clauses x 2 literals, three numeric fields across the rows): each action phase's own condition twice,
# OR-ed, each copy with a numeric literal that always holds - what the widest table shape costs without a change of dynamics
tt_same4 = copy.deepcopy(tt_base)
extras = {2: ("player.total_score >= 0", "player.rounds_as_speaker >= 0"), 3: ("player.lie_index >= 0", "player.total_score <= 255"),
          5: ("player.vote_choice >= 0", "player.rounds_as_speaker <= 15")}
for pid, (e1, e2) in extras.items():
    cc = tt_same4["phases"][str(pid)]["completion_criteria"]
    c0 = cc["target_players"]["condition"]
    cc["target_players"]["condition"] = f"{c0} and {e1} or {c0} and {e2}"
for game, n, rooms, b0, g0, s0, rounds in (("werewolf", 8, 65536, base, gen, same, 1), ("werewolf", 8, 1048576, base, gen, same, 1), ("werewolf", 12, 1048576, base, gen, same, 1),
                                           ("two-truths", 4, 1048576, tt_base, tt_gen, tt_same, 2)):
    res = {}
    for name, dsl in (("shipped", b0), ("generic", g0), ("same", s0)):
        tb = GameTable(dsl, rounds)
        with RoomBatch([(tb, n, rooms)], seed=0xC0FFEE, max_fuse=64, restart=True) as b:
            b.step(256); b.sync()
            b.set_timing(True); b.kernel_time(reset=True)
            b.step(512); b.sync()
            ms, _ = b.kernel_time(reset=True)
            res[name] = ms * 1e3 / 512
    if b0 is tt_base:
        with RoomBatch([(GameTable(tt_same4, rounds), n, rooms)], seed=0xC0FFEE, max_fuse=64, restart=True) as b:
            b.step(256); b.sync()
            b.set_timing(True); b.kernel_time(reset=True)
            b.step(512); b.sync()
            ms, _ = b.kernel_time(reset=True)
        print(f"{game} x{n}, {rooms:>8} rooms: shipped rules in a 2 x 2 generic form (four literal slots, three numeric fields) {ms * 1e3 / 512:7.3f} us/turn (x{ms * 1e3 / 512 / res['shipped']:.2f})", flush=True)
    print(f"{game} x{n}, {rooms:>8} rooms: shipped conditions {res['shipped']:7.3f} us/turn ({rooms / res['shipped'] * 1e6:.3e} steps/s)   "
          f"generic conditions {res['generic']:7.3f} us/turn ({rooms / res['generic'] * 1e6:.3e} steps/s, x{res['generic'] / res['shipped']:.2f})   "
          f"shipped rules in generic form {res['same']:7.3f} us/turn (x{res['same'] / res['shipped']:.2f})", flush=True)
