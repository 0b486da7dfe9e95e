#!/usr/bin/env python3
"""bench.py — room-phase steps/sec of the batch stepper on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], per GPU): 65 536 Werewolf rooms x 8 players,
games/werewolf-(mafia).yaml, all players bots, LLM replaced by the fixed policy, in STEADY STATE
(a finished room is recycled into a new game on its next turn) so that every one of the K timed
steps advances every room through real game logic.  A "step" = one turn (= one LangGraph run in
the reference) of every room of the batch.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task brief) with `roofline` (dominant kernel, HIP
events on the launch stream) and `cpu_baseline` (the oracle's C restatement on the host cores;
the ONLY place bench.py touches oracle/).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ROOMS_PER_GPU = 65536
N_PLAYERS = 8
GAME = "werewolf-(mafia)"
SEED = 0xC0FFEE

# BASELINE.json configs as per-GPU workloads: [(game, players, rooms per GPU)], rooms grouped by game.
# c2 is the contract workload (default); the others are optional extra lines, same JSON format.
WORKLOADS = {
    "c2": [("werewolf-(mafia)", 8, 65536)],
    "c3": [("two-truths-and-a-lie", 4, 1048576)],
    "c4": [("werewolf-(mafia)", 12, 2097152)],                                   # 16 777 216 rooms over 8 GPUs
    "c5": [("werewolf-(mafia)", 8, 524288), ("two-truths-and-a-lie", 4, 524288)],  # 50/50 mix, one launch
}


def load_dsl(game=GAME):
    with open(os.path.join(ROOT, "tests", "golden", "dsl", f"{game}.json"), encoding="utf-8") as f:
        return json.load(f)


def usable_cores():
    """Threads this process may really run at once: the affinity mask, capped by the cgroup CPU quota
    (a GPU box grants a share of its host: oversubscribing 256 hardware threads only adds noise)."""
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
    return n


def cpu_baseline(dsl, budget_s=12.0):
    """The oracle (a scalar C port of the reference loop + policy) on this host's cores, on a
    bounded sample of the same steady-state workload."""
    from oracle.oracle import Oracle
    cores = usable_cores()
    orc = Oracle(dsl, N_PLAYERS)
    rooms = orc.init_rooms(ROOMS_PER_GPU)
    t0 = time.perf_counter()
    orc.run(rooms, SEED, 0, 0, 16, threads=cores, restart=True)        # calibrate
    dt = max(time.perf_counter() - t0, 1e-4)
    turns = int(min(max(16 * budget_s / dt, 64), 4096))
    rooms = orc.init_rooms(ROOMS_PER_GPU)
    t0 = time.perf_counter()
    orc.run(rooms, SEED, 0, 0, turns, threads=cores, restart=True)
    dt_all = time.perf_counter() - t0
    one = orc.init_rooms(4096)
    t1 = max(turns // 8, 16)
    t0 = time.perf_counter()
    orc.run(one, SEED, 0, 0, t1, threads=1, restart=True)
    dt_one = time.perf_counter() - t0
    return {"value": ROOMS_PER_GPU * turns / dt_all, "unit": "room-phase steps/s", "cores": cores,
            "kind": "port",
            "sample": f"{ROOMS_PER_GPU} werewolf x{N_PLAYERS} rooms x {turns} turns, steady state, OpenMP over rooms",
            "single_thread_value": 4096 * t1 / dt_one}


def pmc_traffic(alg_bytes_per_launch):
    """HBM bytes per launch from a committed rocprofv3 --pmc pass of this same command
    (profiles/pmc_traffic.json, written by tools/pmc_summary.py), or None.  The state is read once and
    written once per launch whatever the number of fused turns, so the counters only describe a run
    with the same rooms x record bytes: the file records the launch's read+write floor and is ignored
    when this run's differs."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(p):
        return None
    try:
        with open(p) as f:
            d = json.load(f)
        floor = d.get("state_bytes_read_plus_written")
        if floor is not None and abs(floor - alg_bytes_per_launch) > 1e-6 * alg_bytes_per_launch:
            return None
        return d.get("bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=1024)
    ap.add_argument("--fuse", type=int, default=1024, help="turns fused per launch (1 = one launch per turn)")
    ap.add_argument("--rooms", type=int, default=None, help="rooms per GPU (overrides the workload's count; c2 only)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2", help="BASELINE.json config (default c2 = configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-unfused", action="store_true", help="skip the one-launch-per-turn reference point")
    ap.add_argument("--no-other-shapes", action="store_true", help="skip the informational larger shapes")
    ap.add_argument("--no-from-init", action="store_true", help="skip the S=64-from-initial-state variant")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): everything else any library prints there — RCCL
    # writes a version banner to stdout at communicator creation — is diverted to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from game_engine_amd import GameTable, RoomBatch
    from game_engine_amd.dist import allgather_summary, shard_first_room

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the stepper has no CPU path")
    # one process per GPU; GE_DIST_BACKEND=gloo is only for rehearsing the multi-process flow on a
    # box with fewer GPUs than ranks (ranks then share devices and collectives run on CPU tensors)
    backend = os.environ.get("GE_DIST_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    coll_device = torch.device("cuda", device_index) if backend == "nccl" else torch.device("cpu")
    use_dist = world > 1 or bool(os.environ.get("GE_FORCE_DIST"))      # GE_FORCE_DIST: rehearse RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    dsl = load_dsl()
    table = GameTable(dsl)
    spec = [list(x) for x in WORKLOADS[args.workload]]
    if args.rooms:
        spec[0][2] = args.rooms
    tables = {g: GameTable(load_dsl(g)) for g, _, _ in spec}
    segments = [(tables[g], n, r) for g, n, r in spec]
    rooms = sum(r for _, _, r in spec)
    batch = RoomBatch(segments, seed=SEED, first_room=shard_first_room(rooms, rank),
                      device=device_index, max_fuse=args.fuse, restart=True)
    # algorithmic bytes per room-phase step: record read + written once (room-weighted mean over segments)
    bytes_per_room = sum(batch.bytes_per_room(k) * r for k, (_, _, r) in enumerate(spec)) / rooms
    stream = torch.cuda.current_stream().cuda_stream

    batch.step(args.warmup, stream)                   # untimed; also brings the batch to steady state
    batch.sync()
    batch.set_timing(True)
    batch.kernel_time(reset=True)
    barrier()
    t0 = time.perf_counter()
    batch.step(args.steps, stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = batch.kernel_time(reset=True)
    batch.set_timing(False)

    t_el = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
    if use_dist:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    elapsed = float(t_el.item())

    # the one collective of the path: all-gather of the per-GPU summary (RCCL over xGMI)
    ts = time.perf_counter()
    summary = allgather_summary(batch, world, force=use_dist)
    barrier()
    summary_ms = (time.perf_counter() - ts) * 1e3

    # un-fused reference point: one launch per turn, same workload, short
    unfused = None
    if rank == 0 and world == 1 and not args.no_unfused:      # N=1 only: keeps multi-GPU runs symmetric
        b1 = RoomBatch(segments, seed=SEED, device=device_index, max_fuse=1, restart=True)
        b1.step(256, stream); b1.sync()
        # wall clock first, without per-launch events (they serialise the launches; here the 256
        # launches of a step() call are replayed from a hipGraph), then the kernel time per launch
        t1 = time.perf_counter()
        for _ in range(4):
            b1.step(256, stream)
        b1.sync()
        w1 = (time.perf_counter() - t1) / 4
        b1.set_timing(True); b1.kernel_time(reset=True)
        b1.step(256, stream); b1.sync()
        k1, l1 = b1.kernel_time(reset=True)
        unfused = {"value": rooms * 256 / w1, "ms_per_step": w1 * 1e3 / 256, "kernel_us_per_launch": k1 * 1e3 / max(l1, 1),
                   "achieved_GBs_wall": 2 * bytes_per_room * rooms * 256 / w1 / 1e9,
                   "note": "one launch per turn (max_fuse=1): every turn streams the state through HBM"}
        b1.close()

    # BASELINE.md §3 variant: S = 64 turns from the initial state (no recycling), 3 warm-ups, median of 10
    from_init = None
    if rank == 0 and world == 1 and args.workload == "c2" and not args.no_from_init:
        bi = RoomBatch(segments, seed=SEED, device=device_index, max_fuse=args.fuse, restart=False)
        times = []
        for rep in range(13):
            bi.reset()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            bi.step(64, stream); bi.sync()
            if rep >= 3:
                times.append(time.perf_counter() - t1)
        times.sort()
        med = times[len(times) // 2]
        sm = bi.summary()
        et = bi.read_rooms()["end_turn"]
        live_steps = int(((et >= 0) * (et + 1) + (et < 0) * 64).sum())      # turns a room took before it finished
        from_init = {"value": rooms * 64 / med, "unit": "room-phase steps/s (wall, all rooms, finished ones included)",
                     "live_value": live_steps / med, "live_unit": "room-phase steps/s counting only rooms still in play",
                     "ms_per_64_turns": med * 1e3, "finished_after_64": sm["finished"], "rooms": rooms}
        bi.close()

    # other BASELINE shapes on one GPU (device time, informational; `value` above is the contract number)
    other = None
    if rank == 0 and world == 1 and not args.no_other_shapes and args.workload == "c2":
        other = {}
        tt_dsl = json.load(open(os.path.join(ROOT, "tests", "golden", "dsl", "two-truths-and-a-lie.json"), encoding="utf-8"))
        for label, tb, n, r in (("1048576 Werewolf x8", table, 8, 1 << 20),
                                ("2097152 Werewolf x12 (one GPU's share of C4)", table, 12, 1 << 21),
                                ("1048576 Two-Truths x4 (C3)", GameTable(tt_dsl), 4, 1 << 20)):
            bb = RoomBatch([(tb, n, r)], seed=SEED, device=device_index, max_fuse=args.fuse, restart=True)
            bb.step(128, stream); bb.sync()
            bb.set_timing(True); bb.kernel_time(reset=True)
            bb.step(512, stream); bb.sync()
            ms, nl = bb.kernel_time(reset=True)
            bpr = bb.bytes_per_room(0)
            other[label] = {"value": r * 512 / (ms * 1e-3), "unit": "room-phase steps/s (device time)",
                            "achieved_GBs": 2 * bpr * r * 512 / (ms * 1e-3) / 1e9,
                            "frac": 2 * bpr * r * 512 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_room_record": bpr}
            bb.close()

    if rank == 0:
        total_steps = rooms * world * args.steps
        per_launch_units = rooms * (args.steps / max(launches, 1))
        alg_bytes = 2 * bytes_per_room * per_launch_units
        avg_launch_s = kernel_ms * 1e-3 / max(launches, 1)
        achieved = alg_bytes / avg_launch_s / 1e9
        out = {
            "metric": "room-phase steps/sec", "value": total_steps / elapsed, "unit": "room-phase steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": args.workload + ": " + " + ".join(f"{r} {g} rooms x {n} players" for g, n, r in spec) +
                                   " per GPU, steady state (finished rooms recycled), fixed policy, seed 0xC0FFEE",
                       "rooms_per_gpu": rooms, "n_players": [n for _, n, _ in spec], "turns_fused_per_launch": args.fuse,
                       "bytes_per_room_record": bytes_per_room, "sharding": f"rooms x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak": achieved / 6290.0,
                         "traffic": pmc_traffic(2.0 * bytes_per_room * rooms),
                         "kernel": "ge_step_kernel", "avg_launch_us": avg_launch_s * 1e6, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "algorithmic bytes = 2 x record x rooms x turns in the launch; with fused turns the "
                                 "state stays in registers, so real HBM traffic is ~1/fuse of this (see traffic)"},
            "unfused": unfused,
            "from_init_64": from_init,
            "other_shapes": other,
            "summary": {k: summary[k] for k in ("rooms", "finished", "village_wins", "wolf_wins", "games_recycled", "checksum")},
            "summary_allgather_ms": summary_ms,
        }
        if not args.no_cpu_baseline and args.workload == "c2" and world == 1:   # rank 0 at N=1 only (contract)
            out["cpu_baseline"] = cpu_baseline(dsl)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    batch.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
