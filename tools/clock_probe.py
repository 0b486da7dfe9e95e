#!/usr/bin/env python3
"""The shader clock the fused Werewolf turn loop really runs at, un-profiled (GE_STAMPS=2 build of libge_step.so: every
wavefront reads s_memtime - shader cycles - and s_memrealtime - a constant 100 MHz - at its start and end; ge_kernels.inl).
The issue ceiling bench.py prices against assumes the 2.4 GHz maximum; under a sustained all-SIMD load the chip runs lower
(DVFS, /opt/skills/guides/MI355X_MICROARCH.md "DVFS give-back"), and the vector pipe's busy fraction at the REAL clock is
valu_frac x 2.4 GHz / real clock.
    GE_LIB_PATH=game_engine_amd/ab/sw_clock.so python tools/clock_probe.py [n:rooms ...]     (werewolf only)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "clock_stamps.jsonl")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["GE_STAMPS_OUT"] = out
from game_engine_amd import GameTable, RoomBatch
dsl = json.load(open(os.path.join(ROOT, "tests", "golden", "dsl", "werewolf-(mafia).json"), encoding="utf-8"))
tb = GameTable(dsl)
for spec in sys.argv[1:] or ["8:65536", "8:1048576", "12:2097152"]:
    n, rooms = (int(x) for x in spec.split(":"))
    for fuse, turns in ((1024, 4096), (1, 512)):
        if os.path.exists(out):
            os.remove(out)
        b = RoomBatch([(tb, n, rooms)], seed=0xC0FFEE, max_fuse=fuse, restart=True)
        b.step(1024); b.sync()
        b.set_timing(True); b.kernel_time(reset=True)
        b.step(turns); b.sync()
        ms, launches = b.kernel_time(reset=True)
        b.close()                                                    # the stamps are written when the batch is destroyed
        d = json.loads(open(out).read().strip().splitlines()[-1])
        ghz = d["wave_shader_cycles"] / d["wave_realtime_ticks_100MHz"] * 0.1
        waves = (rooms + 63) // 64
        # SIMD cycles one wave-turn has to itself at the measured clock
        us_turn = ms * 1e3 / turns
        print(f"werewolf x{n} {rooms:>8} rooms fuse {fuse:>4}: {us_turn:8.3f} us/turn, shader clock {ghz:.3f} GHz "
              f"(wave life {d['wave_shader_cycles'] / max(d['wave_turns'], 1):.0f} cycles per wave-turn incl. pre-roll launches)", flush=True)
        if fuse > 1 and os.path.exists(out + ".waves.bin"):
            import numpy as np
            w = np.fromfile(out + ".waves.bin", dtype=np.uint64).reshape(-1, 4)
            w = w[w[:, 1] > 0]
            t0, t1 = int(w[:, 0].min()), int(w[:, 1].max())
            span = (t1 - t0) / 100.0                                     # us
            life = (w[:, 1] - w[:, 0]).astype(np.float64) / 100.0
            hw = w[:, 2]
            simd = (hw >> np.uint64(4)) & np.uint64(3)
            cu = (hw >> np.uint64(8)) & np.uint64(15)
            sh = (hw >> np.uint64(12)) & np.uint64(1)
            se = (hw >> np.uint64(13)) & np.uint64(7)
            xcc = (hw >> np.uint64(32)) & np.uint64(15)
            key = (((xcc * np.uint64(8) + se) * np.uint64(2) + sh) * np.uint64(16) + cu) * np.uint64(4) + simd
            nsimd = len(np.unique(key))
            # residency: waves in flight over time, sampled
            ts = np.linspace(t0, t1, 41)[:-1] + (t1 - t0) / 80.0
            res = [int(((w[:, 0] <= t) & (w[:, 1] > t)).sum()) for t in ts]
            print(f"    last launch: {len(w)} waves on {nsimd} SIMDs, span {span:.0f} us, wave life min / median / max {life.min():.0f} / {np.median(life):.0f} / {life.max():.0f} us, "
                  f"mean resident waves per SIMD {life.sum() / span / nsimd:.2f}; per-SIMD wave counts min / max {np.bincount(np.unique(key, return_inverse=True)[1]).min()} / {np.bincount(np.unique(key, return_inverse=True)[1]).max()}")
            print("    waves in flight per SIMD over the launch (40 samples):", " ".join(f"{r / nsimd:.1f}" for r in res), flush=True)
            starts = np.sort(w[:, 0] - np.uint64(t0)).astype(np.float64) / 100.0
            print("    wave start times (us), deciles:", " ".join(f"{starts[int(q * (len(starts) - 1))]:.0f}" for q in np.linspace(0, 1, 11)), flush=True)

