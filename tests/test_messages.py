"""The human seat at the MESSAGE level (SURVEY §8 a2 + f1; POLICY.md 3b): the browser's strings - vote / input / button / chat /
control (src/app/page.tsx:272-275, 302-305, 341-349, 2774, 2843, 2962) - through `RoomService.handle_message`, against
tests/golden/strings_human_*.json: the reference's AgentState after each message of a scripted person, whose messages the
reference's OWN process_human_action_if_needed logged (oracle/refharness/make_golden.py::string_human_cases).

CPU: the room is stepped by the oracle behind RoomService's batch seam; message classification, resolution, logging and
rendering are the product's.  GPU (-m gpu): the real thing, an N=1 traced batch, in Python and in the Node host."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLD, ROOT, load_dsl, load_golden
from game_engine_amd import messages as M
from test_strings_golden import _check_turn

FILES = sorted(f for f in os.listdir(GOLD) if f.startswith("strings_human_"))


def test_goldens_are_there_and_hold_every_kind_of_message():
    assert len(FILES) >= 3
    kinds, effects, seats = set(), 0, set()
    for name in FILES:
        for case in load_golden(name)["cases"]:
            seats.add(tuple(case["human_seats"]))
            for m in case["messages"]:
                kinds.add((M.classify(m["message"]), m["message"].split(" ")[0]))
                assert m["played"] == (M.classify(m["message"]) != M.CHAT)
                for a in m["actions_added"]:
                    if not a["action"].startswith("[t="):
                        # the reference's own filing: Player 1, phase 0's name, at most 200 characters
                        assert a["player_id"] == "1" and a["phase"] == "Game Introduction" and len(a["action"]) <= 200
                        assert a["action"] == M.logged_text(m["message"])
                        effects += 1
    assert {(M.CHAT, "Player"), (M.CONTROL, "Continue"), (M.CONTROL, "Start"), (M.CONTROL, "I"), (M.ACTION, "Player"),
            (M.ACTION, "Input:"), (M.ACTION, "Button")} <= kinds
    assert effects >= 60 and (1,) in seats and (1, 3) in seats


@pytest.mark.parametrize("text,kind", [
    ("Continue", M.CONTROL), ("  continue ", M.CONTROL), ("Start game.", M.CONTROL), ("START GAME", M.CONTROL),
    ("Player Alice in game chat: hi", M.CHAT), ("Player Alice to Bot 3: hi", M.CHAT),
    ("let us talk to bots", M.CONTROL),                     # lower-case "to bot": not chat-routed (v2:307), not logged (utils.py:338)
    ("Player Alice IN GAME CHAT: hi", M.CONTROL),           # upper case misses the router's test, hits the logger's
    ('Player 1 voted "Bot 2" in voting vote-p7-t11', M.ACTION), ("Input: x", M.ACTION), ("continue please", M.ACTION),
    ('Button "Skip" (ID: b1) has been clicked. Action: skip', M.ACTION), ("", M.ACTION)])
def test_classification(text, kind):
    assert M.classify(text) == kind


@pytest.mark.skipif(not os.path.isdir("/root/reference/agent"), reason="the reference is only in the build container")
def test_classification_against_the_reference_functions():
    """classify / logged_text against the reference's own router test and process_human_action_if_needed."""
    from oracle.refharness.walker import load_reference
    load_reference("v2")
    from langchain_core.messages import HumanMessage
    from tools.utils import process_human_action_if_needed
    dsl = load_dsl("werewolf-(mafia)")
    samples = ["Continue", " continue", "Start game.", "start game", "Player A in game chat: x", "Player A to Bot 2: y", "to bot", "TO BOT 2",
               "Input: " + "z" * 300, 'Player 4 voted "Bot 2" in voting vote-p7-t11', "", "In Game Chat: q", "hello", "CONTINUE"]
    for text in samples:
        logged = process_human_action_if_needed([HumanMessage(content=text)], {}, {}, 0, {"players": [{"gamePlayerId": 1, "name": "Alice"}]}, dsl)
        routed_to_chat = "in game chat:" in text or "to Bot" in text                  # v2:307, quoted
        kind = M.classify(text)
        assert (kind == M.CHAT) == routed_to_chat, text
        if kind == M.ACTION:
            a = logged["1"]["actions"]["1"]
            assert a["action"] == M.logged_text(text) and a["phase"] == dsl["phases"]["0"]["name"] and logged["1"]["name"] == "Alice"
        elif kind == M.CONTROL:
            assert logged == {}, text


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not available")
def test_js_twin_reads_messages_the_same_way():
    """messages.js == messages.py on every message of the goldens and on mangled ones (classification, the 200-character cut
    in code points, and the (seat, choice) readings)."""
    import random
    rnd = random.Random(5)
    texts = {m["message"] for name in FILES for c in load_golden(name)["cases"] for m in c["messages"]}
    texts |= {"Input: " + "\U0001F43A" * 250, "", "Continue ", 'Player 3 voted "Bot 2" in voting vote-p7-t3', 'Player 1 voted "a "b" c" in voting x',
              'Player 1 voted "Bot 2" in voting vote-p7 t3', "Player 2 voted \"Bot\n2\" in voting v", "Player 2 voted \"Bot 2\" in voting v\n"}
    for t in list(texts):
        for _ in range(2):
            k = rnd.randrange(len(t) + 1)
            texts.add(t[:k] + rnd.choice(["", "x", '"', " to Bot ", " TO BOT", "\n"]) + t[k + rnd.randrange(2):])
    texts = sorted(texts)
    names = ["Alice", "Bot 2", "Carol", "Bot 2", "a \"b\" c", "Bot\n2"]
    cases = []
    for t in texts:
        panel = rnd.choice([None, ("vote-p7-t3", ["Bot 2"]), ("x", []), ("v", [])])
        act, pack = rnd.choice([0, 1, 4, 5, 6, 7]), rnd.choice([1, 2])
        alive = [rnd.random() < 0.8 for _ in names]
        seats = rnd.choice([[1], [1, 3], [3, 2], []])
        cases.append({"text": t, "panel": None if panel is None else {"votingId": panel[0], "options": panel[1]}, "act": act, "pack": pack,
                      "names": names, "alive": alive, "seats": seats,
                      "want": [M.classify(t), M.logged_text(t), [list(x) for x in M.resolve(t, panel, act, pack, names, alive, seats)]]})
    js = ("const M=require(process.argv[1]);const cs=JSON.parse(require('fs').readFileSync(0,'utf8'));let bad=[];"
          "cs.forEach((c,i)=>{const got=[M.classify(c.text),M.loggedText(c.text),M.resolve(c.text,c.panel,c.act,c.pack,c.names,c.alive,c.seats)];"
          "if(JSON.stringify(got)!==JSON.stringify(c.want))bad.push([i,got,c.want]);});console.log(JSON.stringify({n:cs.length,bad:bad.slice(0,3)}));")
    out = subprocess.run(["node", "-e", js, os.path.join(ROOT, "game_engine_amd", "node", "messages.js")], input=json.dumps(cases),
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout)
    assert r["n"] == len(cases) and r["bad"] == [], r["bad"]


class _OracleBatch:
    """The subset of RoomBatch a RoomService uses, stepped by the oracle (CPU tests only)."""

    def __init__(self, orc, seed, first_room, human_mask):
        self.orc, self.seed, self.first, self.mask, self.turn = orc, seed, first_room, human_mask, 0
        self.rooms = orc.init_rooms(1)

    def read_rooms(self, first, count):
        from parity_util import oracle_rooms_as_views
        return oracle_rooms_as_views(self.orc, self.rooms).copy()

    def step(self, n):
        assert n == 1
        self.orc.run(self.rooms, self.seed, self.first, self.turn, 1, human_mask=self.mask)
        self.turn += 1

    def read_events(self, first, count):
        from parity_util import oracle_events
        return oracle_events(self.orc, self.rooms, self.turn - 1).reshape(1, 1)

    def inject_action(self, room, player, choice):
        from game_engine_amd.stepper import GeError
        if not self.orc.inject(self.rooms, 0, player, choice):
            raise GeError(-1, "inject_action")

    def close(self):
        pass


def _replay(svc, g, case, where):
    players = [{"name": nm, "gamePlayerId": i + 1, "isBot": (i + 1) not in case["human_seats"]} for i, nm in enumerate(case["names"])]
    svc.create_room("t", g["game"], players, dsl=load_dsl(g["game"]), room_index=case["room"])
    sizes, out = (0, 0, 0), None
    for k, want in enumerate(case["messages"]):
        out = svc.handle_message("t", want["message"])
        assert out["played"] == want["played"], (where, k)
        if not want["played"]:
            assert out["toolCalls"] == [] and out["uiCalls"] == []
        sizes = _check_turn(out["state"], sizes, want, f"{where} message {k}: {want['message'][:60]!r}")
    final = case["final"]
    from test_strings_golden import _strip
    assert _strip(out["state"]["playerActions"]) == final["playerActions"]
    assert out["state"]["game_notes"] == final["game_notes"] and _strip(out["state"]["phase_history"]) == final["phase_history"]
    svc.close()


@pytest.mark.parametrize("name", FILES)
def test_person_messages_equal_reference_run(name):
    from game_engine_amd import RoomService
    from oracle.oracle import Oracle
    g = load_golden(name)
    orc = Oracle(load_dsl(g["game"]), g["n_players"])

    class Svc(RoomService):
        def _new_batch(self, tb, n_players, human_mask, first_room):
            return _OracleBatch(orc, self.seed, first_room, human_mask)

    for case in g["cases"]:
        _replay(Svc(seed=case["seed"]), g, case, f"{name} seed={case['seed']:#x} room={case['room']}")


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_room_service_handle_message_equals_reference_run(name):
    """RoomService.handle_message on an N=1 traced batch on the GPU: the reference run's AgentState after every message."""
    from game_engine_amd import RoomService
    g = load_golden(name)
    for case in g["cases"]:
        _replay(RoomService(seed=case["seed"]), g, case, f"{name} seed={case['seed']:#x} room={case['room']}")


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node is not available")
@pytest.mark.parametrize("name", FILES)
def test_node_room_service_handle_message_equals_reference_run(name):
    g = load_golden(name)
    out = subprocess.run(["node", os.path.join(ROOT, "game_engine_amd", "node", "selftest_messages.js"),
                          os.path.join(GOLD, "dsl", g["game"] + ".json"), os.path.join(GOLD, name)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["ok"] is True and r["messages"] == sum(len(c["messages"]) for c in g["cases"]), r
