#!/usr/bin/env python3
"""Soak run (not collected by pytest; run by hand on a GPU box): many randomly re-conditioned DSLs, player counts, seeds,
batch sizes on both sides of the lone-wavefront / large-batch threshold, fuse patterns, host-driven seats with random
injected actions, and mixed batches of several games in one launch - every room against the oracle after every block of turns.

    python tests/soak_gpu.py [minutes] [seed]

Prints one line per case and a summary; exits non-zero at the first mismatch."""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_dsl                                  # noqa: E402
from game_engine_amd import GameTable, RoomBatch               # noqa: E402
from oracle import cond_gen, dsl_table as T, dsl_variants      # noqa: E402
from oracle.oracle import Oracle                               # noqa: E402
from parity_util import assert_views_equal, oracle_rooms_as_views   # noqa: E402

GAMES = {1: "werewolf-(mafia)", 2: "two-truths-and-a-lie"}
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
t_end = time.time() + 60.0 * minutes
cases = rooms_turns = 0
def mixed_case():
    """a mixed batch: 2 - 4 segments (games, player counts, host-driven seats of their own) advanced by the same launches;
    global room indices run across the segments, every segment against its own oracle"""
    global rooms_turns
    k = rng.randint(2, 4)
    seed, first = rng.randrange(1 << 48), rng.randrange(1 << 40)
    restart, max_fuse = rng.random() < 0.8, rng.choice([0, 0, 1, 3, 16, 64])
    segs = []
    for _ in range(k):
        pack = rng.choice([1, 2])
        if rng.random() < 0.3:                                 # a reference-run rule variant of the game
            name = rng.choice(sorted(v for v in dsl_variants.VARIANTS if dsl_variants.VARIANTS[v][0].startswith("werewolf") == (pack == 1)) or [None])
        else:
            name = None
        rounds = 1
        if name:
            game, builder, rounds = dsl_variants.VARIANTS[name]
            dsl = builder(load_dsl(game))
        else:
            dsl = load_dsl(GAMES[pack])
        n = rng.randint(4, 12) if pack == 1 else rng.randint(3, 12)
        if pack == 2 and rounds * 2 * (n - 1) > 255:
            rounds = 1
        R = rng.choice([1, 63, 64, 65, 255, 257, 1000, 4097, 20000, 70000])
        mask = rng.choice([0, 0, 0, 1, 1 << (n - 1), 0b101]) & ((1 << n) - 1)
        orc = Oracle(dsl, n, rounds=rounds)
        segs.append(dict(dsl=dsl, n=n, R=R, mask=mask, rounds=rounds, orc=orc, want=orc.init_rooms(R)))
    t = 0
    with RoomBatch([(GameTable(g["dsl"], g["rounds"]), g["n"], g["R"], g["mask"]) for g in segs], seed=seed, first_room=first,
                   restart=restart, max_fuse=max_fuse) as b:
        for block in range(rng.randint(2, 4)):
            turns = rng.choice([1, 2, 7, 33, 64, 100])
            b.step(turns)
            got, lo = b.read_rooms(), 0
            for g in segs:
                g["orc"].run(g["want"], seed, first + lo, t, turns, threads=0, restart=restart, human_mask=g["mask"])
                assert_views_equal(got[lo:lo + g["R"]], oracle_rooms_as_views(g["orc"], g["want"]),
                                   f"mixed segment n={g['n']} R={g['R']} at {lo} seed={seed} fuse={max_fuse} t={t + turns}")
                lo += g["R"]
            t += turns
    rooms_turns += sum(g["R"] for g in segs) * t
    print("ok  mixed batch        " + " + ".join(f"{g['R']}x{g['n']}" for g in segs) + f"  fuse={max_fuse:<2} restart={int(restart)} turns={t}", flush=True)


while time.time() < t_end:
    if rng.random() < 0.15:
        mixed_case()
        cases += 1
        continue
    pack = rng.choice([1, 1, 2])
    kind = rng.random()
    rounds = 1
    if kind < 0.55:                                            # random re-conditioning of the game (generic builds mostly)
        base = load_dsl(GAMES[pack])
        acts = {p.id: p.act for p in T.compile_dsl(base).phases}
        dsl = cond_gen.randomize_dsl(random.Random(rng.randrange(1 << 30)), base, pack, acts)
        try:
            T.compile_dsl(dsl)
        except T.DslError:
            continue
        what = "random conditions"
    elif kind < 0.75:
        name = rng.choice(sorted(dsl_variants.VARIANTS))
        game, builder, rounds = dsl_variants.VARIANTS[name]
        pack = 1 if game.startswith("werewolf") else 2
        dsl = builder(load_dsl(game))
        what = name
    elif kind < 0.85 and pack == 1:
        dsl, what = load_dsl("draft-werewolf-(mafia)"), "draft DSL"
    else:
        dsl, what = load_dsl(GAMES[pack]), "shipped"
    n = rng.randint(4, 12) if pack == 1 else rng.randint(3, 12)
    if pack == 2 and rounds * 2 * (n - 1) > 255:
        rounds = 1
    R = rng.choice([1, 63, 64, 65, 1000, 4097, 20000, 65536, 65537, 70000, 150000, 300000])
    seed, first = rng.randrange(1 << 48), rng.randrange(1 << 40)
    mask = rng.choice([0, 0, 0, 1, 1 << (n - 1), 0b101]) & ((1 << n) - 1)
    restart = rng.random() < 0.8
    max_fuse = rng.choice([0, 0, 1, 3, 16, 64])
    orc = Oracle(dsl, n, rounds=rounds)
    want = orc.init_rooms(R)
    humans = [i + 1 for i in range(n) if (mask >> i) & 1]
    nprng = np.random.default_rng(rng.randrange(1 << 30))
    t = 0
    with RoomBatch([(GameTable(dsl, rounds), n, R, mask)], seed=seed, first_room=first, restart=restart, max_fuse=max_fuse) as b:
        for block in range(rng.randint(2, 5)):
            if humans and R <= 20000:                          # random host-driven actions, accepted or refused alike
                k = min(R, 500)
                rr = nprng.integers(0, R, size=k).astype(np.uint64)
                pl = nprng.choice(humans, size=k).astype(np.uint32)
                ch = nprng.integers(0, n + 2, size=k).astype(np.uint32)
                exp = np.array([0 if orc.inject(want, int(r), int(p), int(c)) else -1 for r, p, c in zip(rr, pl, ch)], dtype=np.int32)
                got = b.inject_actions(rr, pl, ch)
                assert ((got == 0) == (exp == 0)).all(), ("inject", what, n, R)
            turns = rng.choice([1, 2, 7, 33, 64, 100])
            b.step(turns)
            orc.run(want, seed, first, t, turns, threads=0, restart=restart, human_mask=mask)
            t += turns
            assert_views_equal(b.read_rooms(), oracle_rooms_as_views(orc, want), f"{what} pack={pack} n={n} R={R} seed={seed} fuse={max_fuse} t={t}")
        rooms_turns += R * t
    cases += 1
    print(f"ok  {what:<18} pack={pack} n={n:<2} rooms={R:<6} fuse={max_fuse:<2} restart={int(restart)} humans={mask:#x} turns={t}", flush=True)
print(f"soak: {cases} cases, {rooms_turns:.3e} room-turns compared room by room, no mismatch")
