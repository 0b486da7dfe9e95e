"""Pins the oracle (oracle/ge_oracle.c via oracle/oracle.py) against golden vectors that were
produced by running the REFERENCE's own node coroutines (game_agent_v2.py / v3) under the
fixed policy — turn by turn, bit-exact.  CPU only."""
import pytest

from conftest import golden_dsl, golden_files, human_files, load_dsl, load_golden, restart_files
from oracle.oracle import Oracle


@pytest.mark.parametrize("name", golden_files())
def test_oracle_matches_reference_trajectories(name):
    g = load_golden(name)
    orc = Oracle(golden_dsl(g), g["n_players"], rounds=g["rounds"])
    for case in g["cases"]:
        got = orc.trajectory(case["seed"], case["room"], len(case["turns"]))
        for t, (a, b) in enumerate(zip(got, case["turns"])):
            assert a == b, f"{name} seed={case['seed']:#x} room={case['room']} turn={t}"


def test_oracle_batched_equals_stepwise(dsl_ww):
    """orc_run over many rooms/turns == per-turn stepping; threads do not change results."""
    import numpy as np
    orc = Oracle(dsl_ww, 8)
    a = orc.init_rooms(257)
    b = orc.init_rooms(257)
    orc.run(a, 0xC0FFEE, 1000, 0, 64, threads=1)
    for t in range(64):
        orc.run(b, 0xC0FFEE, 1000, t, 1, threads=4)
    assert a.tobytes() == b.tobytes()
    assert (a["end_turn"] >= 0).mean() > 0.5


@pytest.mark.parametrize("name", restart_files())
def test_oracle_restart_mode_matches_chained_reference_sessions(name):
    """Steady-state mode: a finished room is replaced, on its next turn, by a new reference
    session whose clock starts there."""
    g = load_golden(name)
    orc = Oracle(golden_dsl(g), g["n_players"])
    for case in g["cases"]:
        got = orc.trajectory(case["seed"], case["room"], len(case["turns"]), restart=True)
        for t, (a, b) in enumerate(zip(got, case["turns"])):
            assert a == b, f"{name} seed={case['seed']:#x} turn={t}"


@pytest.mark.parametrize("name", human_files())
def test_oracle_with_host_driven_player_matches_reference(name):
    """Player 1 is the reference's human: the bot policy skips it, a scripted person acts for it
    (oracle/human_script.py) and its action is logged at the start of the next graph run."""
    from oracle.human_script import scripted_human
    g = load_golden(name)
    orc = Oracle(golden_dsl(g), g["n_players"])
    n = g["n_players"]
    acted_as_human = 0
    for case in g["cases"]:
        def human(t, proj):
            nonlocal acted_as_human
            a = scripted_human(orc.table, t, proj, n)
            acted_as_human += a is not None
            return a
        got = orc.trajectory(case["seed"], case["room"], len(case["turns"]), human_mask=g["human_mask"], human=human)
        for t, (a, b) in enumerate(zip(got, case["turns"])):
            assert a == b, f"{name} seed={case['seed']:#x} turn={t}"
    assert acted_as_human >= 3          # the script really drove player 1
