"""Times the REFERENCE's Python loop (v2 / v3 nodes) under FixedPolicy on this container's CPU:
BASELINE.json config C1 (1 Werewolf room x 8 bots, LLM stubbed).  Build container only.
    python -m oracle.refharness.time_reference"""
import statistics
import time

from .walker import RoomSession


def run(version, turns=300, reps=5):
    rates = []
    for rep in range(reps):
        n, t0, sess = 0, None, None
        while n < turns + 1:
            if sess is None or sess.end_turn >= 0:          # chain games like the steady-state mode
                sess = RoomSession("werewolf-(mafia)", 8, seed=rep, room=0, version=version, turn0=n)
            if n == 1:
                t0 = time.perf_counter()                    # first turn (DSL load) is warm-up
            sess.step()
            n += 1
        rates.append(turns / (time.perf_counter() - t0))
    return statistics.median(rates)


def _one(version):
    return run(version, reps=3)


if __name__ == "__main__":
    import multiprocessing as mp
    import sys
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for v in ("v2", "v3"):
        if procs == 1:
            r = run(v)
            print(f"{v}: {1e3 / r:.3f} ms/turn  {r:.0f} room-phase steps/s (1 core)")
        else:                                   # independent rooms, one process each
            with mp.get_context("spawn").Pool(procs) as pool:
                rates = pool.map(_one, [v] * procs)
            print(f"{v}: {sum(rates):.0f} room-phase steps/s over {procs} processes ({min(rates):.0f}..{max(rates):.0f} each)")
