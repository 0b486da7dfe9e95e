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


if __name__ == "__main__":
    for v in ("v2", "v3"):
        r = run(v)
        print(f"{v}: {1e3 / r:.3f} ms/turn  {r:.0f} room-phase steps/s (1 core)")
