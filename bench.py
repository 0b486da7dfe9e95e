#!/usr/bin/env python3
"""bench.py — room-phase steps/sec of the batch stepper on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 without WORLD_SIZE in the environment: this process starts N fresh rank processes
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`) BEFORE anything touches the GPU
and relays rank 0's JSON line.  Launched under torch.distributed.run it is one rank of N.

Workload (BASELINE.json configs[1], per GPU): 65 536 Werewolf rooms x 8 players,
games/werewolf-(mafia).yaml, all players bots, LLM replaced by the fixed policy, in STEADY STATE:
a finished room is recycled into a new game on its next turn, and the batch is always pre-rolled
(untimed, >= PREROLL_TURNS turns, whatever --warmup says) so that the timed turns see rooms spread
over all phases of the game.

One bench STEP = one pass of the hot path over the whole batch = ONE fused launch: every room record
is loaded from HBM once, advanced by `turns_fused_per_launch` turns (= LangGraph runs of the
reference) in registers, and stored once.  `value` counts room-phase steps (room-turns):
    value = rooms x ranks x turns_fused_per_launch x K / time of the K launches.
Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task brief):
  roofline        the contract's block for the dominant kernel: SURVEY 8(d)'s ALGORITHMIC bytes per launch / launch
                  duration (HIP events on the launch stream) against the HBM peak.  With fused turns that is a yardstick,
                  not traffic - `bound_actual` names the bound that really holds (instruction issue) and
                  `hbm_frac_of_measured_traffic` what the measured bytes amount to;
  issue           wave-instructions per second against the SIMD issue ceiling (counters of the committed PMC pass of
                  this same shape and fuse; dropped when the kernel sources have changed since);
  hbm_streaming   the max_fuse = 1 launch, where every turn reads and writes every record: a memory-side rate for the shapes
                  whose state fits the 256 MiB Infinity Cache (the contract shape and the BASELINE shapes do);
  hbm_streaming_beyond_l3  (N = 1) the same launch over 640 MiB - 1 GiB of resident state: the HBM fraction that is provably
                  HBM, with a parity check (single-turn launches == fused turns, summary checksum) at that size;
  other_shapes    (N = 1) the other BASELINE shapes on this GPU, each with its own fused figure, hbm_streaming block
                  and a bounded CPU row;
  other_workloads (N > 1) BASELINE configs[3] and [4]: every rank steps its share of C4 (16 777 216 Werewolf x 12 over 8
                  GPUs = 2 097 152 per rank) and C5 (50/50 mix), with the timed summary all-gather, each with its own
                  roofline block, and C4 with the single-turn launches of every rank's share (hbm_streaming);
  cpu_baseline    the oracle's C restatement on the host cores of rank 0 (the ONLY place bench.py touches oracle/).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md:36
HBM_COPY_GBS = 6290.0          # measured float4 copy, same line
N_SIMD = 1024                  # 256 CUs x 4 SIMD-32
CLOCK_HZ = 2.4e9               # max shader clock (guide :34)
PREROLL_TURNS = 1024           # a Werewolf x8 game lasts ~40 turns: >= 25 games per slot before timing
ROOMS_PER_GPU = 65536
N_PLAYERS = 8
GAME = "werewolf-(mafia)"
SEED = 0xC0FFEE
WW, TT = "werewolf-(mafia)", "two-truths-and-a-lie"

# BASELINE.json configs as per-GPU workloads: [(game, players, rooms per GPU)], rooms grouped by game.
# c2 is the contract workload (default); the others are optional extra lines, same JSON format.
WORKLOADS = {
    "c2": [(WW, 8, 65536)],
    "c3": [(TT, 4, 1048576)],
    "c4": [(WW, 12, 2097152)],                                   # 16 777 216 rooms over 8 GPUs
    "c5": [(WW, 8, 524288), (TT, 4, 524288)],                    # 50/50 mix, one launch
    # resident states larger than the 256 MiB Infinity Cache (the hbm_streaming_beyond_l3 shapes; `--workload` lets
    # tools/profile.sh take their kernel-trace and counter passes)
    "ww8_33554432": [(WW, 8, 1 << 25)],
    "c4_whole": [(WW, 12, 1 << 24)],                             # the whole of C4 on one GPU
    "tt4_33554432": [(TT, 4, 1 << 25)],
}
L3_BYTES = 256 << 20           # Infinity Cache (MALL) of MI355X, /opt/skills/guides/MI355X_MICROARCH.md:297
# single-turn launches over a resident state LARGER than the Infinity Cache: what a launch reads and writes cannot have been
# left in the cache by the launch before it - the HBM figure that is provably HBM (label, profile key, segments)
BEYOND_L3_SHAPES = (
    ("33554432 Werewolf x8 (1 GiB of records)", "ww8_33554432"),
    ("16777216 Werewolf x12 (the WHOLE of C4 on one GPU, 640 MiB)", "c4_whole"),
    ("33554432 Two-Truths x4 (768 MiB)", "tt4_33554432"),
)
# the informational shapes of an N = 1 run: (label, profile key, per-GPU segments)
OTHER_SHAPES = (
    ("1048576 Werewolf x8", "ww8_1048576", [(WW, 8, 1 << 20)]),
    ("2097152 Werewolf x12 (one GPU's share of C4)", "c4", WORKLOADS["c4"]),
    ("1048576 Two-Truths x4 (C3)", "c3", WORKLOADS["c3"]),
    ("524288 Werewolf x8 + 524288 Two-Truths x4 (one GPU's share of C5)", "c5", WORKLOADS["c5"]),
)


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


def cpu_baseline(spec, budget_s=12.0, sample_rooms=None, single_thread=True):
    """The oracle (a scalar C port of the reference loop + policy) on this host's cores, on a bounded sample of the
    same steady-state workload: every segment's game in the workload's own proportions, `sample_rooms` rooms in all
    (default: the workload's own count, capped at 1 048 576), as many turns as fit the time budget."""
    from oracle.oracle import Oracle
    total = sum(r for _, _, r in spec)
    want = min(sample_rooms or total, 1 << 20)
    cores = usable_cores()
    parts = [(Oracle(load_dsl(g), n), g, n, max(1, want * r // total)) for g, n, r in spec]
    t0 = time.perf_counter()
    for orc, _, _, rooms in parts:                                   # calibrate
        orc.run(orc.init_rooms(rooms), SEED, 0, 0, 8, threads=cores, restart=True)
    dt = max(time.perf_counter() - t0, 1e-4)
    turns = int(min(max(8 * budget_s / dt, 16), 16384))
    states = [orc.init_rooms(rooms) for orc, _, _, rooms in parts]
    t0 = time.perf_counter()
    for (orc, _, _, _), state in zip(parts, states):
        orc.run(state, SEED, 0, 0, turns, threads=cores, restart=True)
    dt_all = time.perf_counter() - t0
    rooms_all = sum(p[3] for p in parts)
    out = {"value": rooms_all * turns / dt_all, "unit": "room-phase steps/s", "cores": cores, "kind": "port",
           "sample": " + ".join(f"{rooms} {g} x{n}" for _, g, n, rooms in parts) + f" rooms x {turns} turns, steady state, OpenMP over rooms"}
    if single_thread:
        orc = parts[0][0]
        one = orc.init_rooms(4096)
        t1 = max(turns // 8, 16)
        t0 = time.perf_counter()
        orc.run(one, SEED, 0, 0, t1, threads=1, restart=True)
        out["single_thread_value"] = 4096 * t1 / (time.perf_counter() - t0)
    return out


def committed_profile(key, fuse):
    """Counters of a committed rocprofv3 --pmc run of this same shape AND fuse setting (profiles/pmc_<key>.json for fused
    launches, pmc_<key>_k1.json for --fuse 1; written by tools/pmc_summary.py), or {}.  PMC counters cannot be collected
    from inside the timed run, so the bench line quotes the committed passes and says so (`source`).  A profile taken
    on other kernel sources than the ones this library was built from is stale: its instruction counts are not quoted."""
    from game_engine_amd._lib import kernel_source_hash
    p = os.path.join(ROOT, "profiles", f"pmc_{key}_k1.json" if fuse == 1 else f"pmc_{key}.json")
    try:
        with open(p) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return {}
    if fuse != 1 and int(d.get("turns_per_launch", fuse)) == 1:
        return {}
    d["_path"] = os.path.relpath(p, ROOT)
    d["_stale"] = d.get("kernel_src_sha256") != kernel_source_hash()
    return d


def spawn_ranks(args, argv):
    """--gpus N > 1 and no WORLD_SIZE: start N fresh rank processes.  The parent has imported neither
    torch nor the HIP library at this point, so nothing here has touched the GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in p.stdout.decode("utf-8", "replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if p.returncode != 0 or line is None:
        sys.stderr.write(p.stdout.decode("utf-8", "replace"))
        raise SystemExit(p.returncode or 1)
    sys.stdout.write(line + "\n")
    sys.stdout.flush()


def valu_price(key):
    """(mean cycles per vector instruction of this shape's fused kernel, cycles per instruction for a lone wavefront) from
    profiles/valu_mix.json (tools/valu_mix.py --write: the binary's instruction mix priced by tools/microbench/encoding_probe.hip's
    per-class costs), or None when it was made for other device code"""
    from game_engine_amd._lib import kernel_source_hash
    try:
        with open(os.path.join(ROOT, "profiles", "valu_mix.json")) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    if d.get("kernel_src_sha256") != kernel_source_hash() or key not in d.get("mean_price_cycles", {}):
        return None
    return float(d["mean_price_cycles"][key]), float(d.get("lone_wavefront_cycles", 5.2))


def issue_block(prof, rooms, turns, kernel_s, key=None):
    """The bound of the fused kernel: wave-instructions issued per second against the SIMD issue
    ceiling.  A wave64 VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md:54, :473)
    -> 1 024 SIMDs x 2.4 GHz / 2 = 1.23e12 wave-instr/s with >= 2 wavefronts per SIMD; a wavefront ALONE
    on its SIMD issues one instruction per 4 cycles (same table, row 'vector-instruction ISSUE cost')
    -> 6.1e11.  Instructions per wave-turn come from the committed SQ counter pass (PMC cannot be
    collected inside the timed run)."""
    waves = (rooms + 63) // 64
    waves_per_simd = waves / N_SIMD
    cyc = 4.0 if waves_per_simd < 2.0 else 2.0
    ceiling = N_SIMD * CLOCK_HZ / cyc
    wave_turns_per_s = waves * turns / kernel_s
    out = {"bound": "valu-issue", "waves_per_simd": waves_per_simd, "cycles_per_wave_instruction": cyc,
           "ceiling_wave_instr_per_s": ceiling, "wave_turns_per_s": wave_turns_per_s,
           # SIMD cycles (at 2.4 GHz) one wave-turn has to itself: busy SIMDs x clock / wave-turns per second
           "simd_cycles_per_wave_turn": min(waves, N_SIMD) * CLOCK_HZ / wave_turns_per_s}
    ipt = prof.get("instructions_per_wave_turn")
    if ipt and not prof.get("_stale"):
        total = float(ipt.get("valu", 0)) + float(ipt.get("salu", 0)) + float(ipt.get("lds", 0))
        out.update({"instructions_per_wave_turn": ipt, "wave_instr_per_s": total * wave_turns_per_s,
                    "frac": total * wave_turns_per_s / ceiling,
                    "valu_frac": float(ipt.get("valu", 0)) * wave_turns_per_s / ceiling,
                    "wait_any_frac_of_wave_cycles": prof.get("wait_any_frac"),
                    "source": f"{prof.get('_path')} (committed rocprofv3 --pmc SQ pass of this shape and fuse setting, same kernel sources)"})
        # the same count at what the instructions really cost a SIMD (profiles/r05_encoding_probe.txt: 2.3 cycles for a simple op,
        # 4.15 for fused / bit-field / multiply / compare / DPP ...; a lone wavefront ~5.2 per instruction of any kind): the share of
        # its SIMD's cycles a wave-turn's vector instructions occupy.  valu_frac (nominal 2 cycles) FALLS when cheap instructions are removed
        price = valu_price(key) if (key and turns > 1) else None
        if price:
            cyc_i = price[1] if waves_per_simd < 2.0 else price[0]
            out.update({"valu_mean_price_cycles": cyc_i,
                        "valu_priced_frac": float(ipt.get("valu", 0)) * cyc_i / out["simd_cycles_per_wave_turn"],
                        "valu_price_source": "profiles/valu_mix.json (tools/valu_mix.py: static instruction mix x tools/microbench/encoding_probe.hip class costs, same kernel sources)"})
    else:
        out.update({"instructions_per_wave_turn": None, "frac": None,
                    "source": (f"{prof.get('_path')} is stale: taken on other kernel sources (kernel_src_sha256 differs)" if ipt else None)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16, help="timed steps; one step = one fused launch of --fuse turns of every room")
    ap.add_argument("--warmup", type=int, default=2, help="untimed steps (launches) before the timed region, after the fixed pre-roll")
    ap.add_argument("--fuse", type=int, default=1024, help="turns fused per launch = turns per bench step")
    ap.add_argument("--rooms", type=int, default=None, help="rooms per GPU (overrides the workload's count; c2 only)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2", help="BASELINE.json config (default c2 = configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-unfused", action="store_true", help="skip the one-launch-per-turn points")
    ap.add_argument("--no-other-shapes", action="store_true", help="skip the other BASELINE shapes (N = 1) / workloads (N > 1)")
    ap.add_argument("--no-from-init", action="store_true", help="skip the S=64-from-initial-state variant")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0 or args.fuse < 1:
        raise SystemExit("bench.py: --gpus/--steps/--fuse must be >= 1, --warmup >= 0")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # stdout carries exactly ONE line (the JSON): everything else any library prints there — RCCL
    # writes a version banner to stdout at communicator creation — is diverted to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from game_engine_amd import GameTable, RoomBatch
    from game_engine_amd.dist import allgather_summary, shard_first_room

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the stepper has no CPU path")
    # one process per GPU; GE_DIST_BACKEND=gloo is only for rehearsing the multi-process flow on a
    # box with fewer GPUs than ranks (ranks then share devices and collectives run on CPU tensors)
    backend = os.environ.get("GE_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py: {world} ranks but {n_dev} GPUs (one process per GPU; GE_DIST_BACKEND=gloo rehearses with shared devices)")
    device_index = local_rank % n_dev
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

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=coll_device)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    tables = {}

    def segments_of(spec):
        for g, _, _ in spec:
            if g not in tables:
                tables[g] = GameTable(load_dsl(g))
        return [(tables[g], n, r) for g, n, r in spec]

    def record_bytes(batch, spec):
        """algorithmic bytes per room-phase step / 2: the record, room-weighted mean over the segments"""
        rooms = sum(r for _, _, r in spec)
        return sum(batch.bytes_per_room(k) * r for k, (_, _, r) in enumerate(spec)) / rooms

    stream = torch.cuda.current_stream().cuda_stream

    def streaming_point(spec, launches=256, first_room=0, preroll=PREROLL_TURNS, parity=True):
        """The HBM-streaming point of a shape: max_fuse = 1, one launch per turn - every turn reads and writes every record
        through HBM.  Device time per launch by HIP events on the launch stream, and the wall clock of hipGraph replays."""
        segs = segments_of(spec)
        rooms = sum(r for _, _, r in spec)
        b1 = RoomBatch(segs, seed=SEED, first_room=first_room, device=device_index, max_fuse=1, restart=True)
        bpr = record_bytes(b1, spec)
        b1.step(preroll, stream); b1.step(launches, stream); b1.sync()   # pre-roll; the second call builds the graph of `launches` launches
        # wall clock first, without per-launch events (they serialise the launches; here the launches of a step() call are
        # replayed from a hipGraph), then the kernel time per launch
        t1 = time.perf_counter()
        for _ in range(4):
            b1.step(launches, stream)
        b1.sync()
        w1 = (time.perf_counter() - t1) / 4
        # device time of one replay: HIP events around the whole graph on the launch stream (no host latency in it, and no
        # per-launch event overhead: what back-to-back single-turn launches sustain)
        reps = []
        for _ in range(3):                                         # the median of three replays (a single one moves by a few per cent)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); b1.step(launches, stream); e1.record(); e1.synchronize()
            reps.append(e0.elapsed_time(e1) * 1e-3)
        g1 = sorted(reps)[1]
        b1.set_timing(True); b1.kernel_time(reset=True)
        b1.step(launches, stream); b1.sync()
        k1, l1 = b1.kernel_time(reset=True)
        b1.close()
        us = k1 * 1e3 / max(l1, 1)
        state_bytes = bpr * rooms
        beyond = state_bytes > L3_BYTES
        gbs_wall = 2 * bpr * rooms * launches / w1 / 1e9
        gbs_kernel = 2 * bpr * rooms / (us * 1e-6) / 1e9
        gbs_graph = 2 * bpr * rooms * launches / g1 / 1e9
        pt = {"value": rooms * launches / w1, "unit": "room-phase steps/s (wall)", "ms_per_turn": w1 * 1e3 / launches,
              "bound": "hbm", "bytes_per_launch": 2 * bpr * rooms, "rooms": rooms, "launches_per_replay": launches,
              "resident_state_MiB": state_bytes / (1 << 20), "fits_infinity_cache": not beyond,
              # what `frac` is: state > 256 MiB Infinity Cache -> at least hbm_floor_frac of it is DRAM traffic (a launch cannot
              # read more than 256 MiB of what the previous one left in the cache); else a memory-side rate the cache may serve
              "what_frac_is": "hbm (state > Infinity Cache)" if beyond else "memory-side (state fits the Infinity Cache)",
              # sustained: device time of a replayed graph of back-to-back launches / launches
              "us_per_launch_sustained": g1 * 1e6 / launches, "achieved_GBs": gbs_graph, "frac": gbs_graph / HBM_PEAK_GBS,
              "hbm_floor_frac": (gbs_graph / HBM_PEAK_GBS) * max(0.0, 1.0 - L3_BYTES / state_bytes),
              "frac_of_measured_copy_peak": gbs_graph / HBM_COPY_GBS,
              # one launch at a time between two HIP events (includes ~2 us of event / dispatch gap per launch)
              "kernel_us_per_launch": us, "frac_kernel": gbs_kernel / HBM_PEAK_GBS, "frac_wall": gbs_wall / HBM_PEAK_GBS,
              "launch_chains": os.environ.get("GE_CHAINS", "default")}
        if parity:
            with RoomBatch(segs, seed=SEED, first_room=first_room, device=device_index, max_fuse=1, restart=True) as k1b, \
                 RoomBatch(segs, seed=SEED, first_room=first_room, device=device_index, max_fuse=64, restart=True) as fz:
                k1b.step(64, stream); fz.step(64, stream)
                s1, sf = k1b.summary(), fz.summary()
            pt["parity"] = {"turns": 64, "checksum_single_turn": s1["checksum"], "checksum_fused": sf["checksum"],
                            "single_turn_equals_fused": s1 == sf}
        return pt

    spec = [list(x) for x in WORKLOADS[args.workload]]
    if args.rooms:
        spec[0][2] = args.rooms
    spec = [tuple(x) for x in spec]
    segments = segments_of(spec)
    rooms = sum(r for _, _, r in spec)
    batch = RoomBatch(segments, seed=SEED, first_room=shard_first_room(rooms, rank),
                      device=device_index, max_fuse=args.fuse, restart=True)
    bytes_per_room = record_bytes(batch, spec)

    # untimed: fixed pre-roll to steady state, then W warm-up steps of the timed shape
    preroll = 0
    while preroll < PREROLL_TURNS:
        batch.step(args.fuse, stream)
        preroll += args.fuse
    for _ in range(args.warmup):
        batch.step(args.fuse, stream)
    batch.sync()
    batch.set_timing(True)
    batch.kernel_time(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):                       # K steps = K fused launches
        batch.step(args.fuse, stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = batch.kernel_time(reset=True)
    batch.set_timing(False)
    elapsed = max_over_ranks(elapsed)

    # the one collective of the path: all-gather of the per-GPU summary (RCCL over xGMI)
    ts = time.perf_counter()
    summary = allgather_summary(batch, world, force=use_dist)
    barrier()
    summary_ms = (time.perf_counter() - ts) * 1e3

    # the HBM-streaming point of the contract shape (N=1 only: keeps multi-GPU runs symmetric)
    unfused = None
    if rank == 0 and world == 1 and not args.no_unfused:
        unfused = streaming_point(spec, launches=max(16, min(256, int(1.2e10 // rooms))), preroll=PREROLL_TURNS if rooms <= (1 << 22) else 256)
        unfused["shape"] = args.workload

    # the same launch over a state larger than the Infinity Cache (N = 1), and parity at that size: 64 single-turn launches
    # == 64 fused turns (the whole ge_summary, checksum of every packed record included)
    beyond_l3 = None
    if rank == 0 and world == 1 and not args.no_unfused and not args.no_other_shapes and args.workload == "c2":
        beyond_l3 = {}
        for label, key in BEYOND_L3_SHAPES:
            sp = WORKLOADS[key]
            r = sum(x[2] for x in sp)
            pt = streaming_point(sp, launches=max(8, min(64, int(3e9 // r))), preroll=256)
            pt["profile"] = f"profiles/r05_{key}_k1_kernel_stats.csv, profiles/pmc_{key}_k1.json"
            beyond_l3[label] = pt

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

    # N = 1: the other BASELINE shapes on this GPU (device time; `value` above is the contract number), each with the
    # launch that really streams it through HBM and a bounded CPU row
    other = None
    if rank == 0 and world == 1 and not args.no_other_shapes and args.workload == "c2":
        other = {}
        for label, key, sp in OTHER_SHAPES:
            r = sum(x[2] for x in sp)
            bb = RoomBatch(segments_of(sp), seed=SEED, device=device_index, max_fuse=args.fuse, restart=True)
            turns = 2 * args.fuse                                  # two launches of the contract's fuse setting, after one as pre-roll
            bb.step(args.fuse, stream); bb.sync()
            bb.set_timing(True); bb.kernel_time(reset=True)
            bb.step(turns, stream); bb.sync()
            ms, nl = bb.kernel_time(reset=True)
            bpr = record_bytes(bb, sp)
            bb.close()
            alg = 2 * bpr * r * turns / (ms * 1e-3) / 1e9
            other[label] = {"value": r * turns / (ms * 1e-3), "unit": "room-phase steps/s (device time)",
                            "us_per_turn": ms * 1e3 / turns, "turns_timed": turns, "bytes_per_room_record": bpr,
                            "algorithmic_GBs": alg, "algorithmic_frac": alg / HBM_PEAK_GBS,
                            "bound_actual": "valu-issue",
                            "issue": issue_block(committed_profile(key, args.fuse), r, turns, ms * 1e-3, key if args.fuse > 1 else None),
                            "hbm_streaming": None if args.no_unfused else streaming_point(sp, launches=128)}
            if not args.no_cpu_baseline:
                other[label]["cpu_baseline"] = cpu_baseline(sp, budget_s=4.0, sample_rooms=1 << 18, single_thread=False)

    # N > 1: BASELINE configs[3] (C4) and configs[4] (C5) - every rank steps its share, no data-path collective; the
    # summary all-gather is timed on its own
    other_workloads = None
    if world > 1 and not args.no_other_shapes and args.workload == "c2":
        other_workloads = {}
        k_timed = max(2, min(args.steps, 8))
        for key in ("c4", "c5"):
            sp = WORKLOADS[key]
            r = sum(x[2] for x in sp)
            bb = RoomBatch(segments_of(sp), seed=SEED, first_room=shard_first_room(r, rank), device=device_index,
                           max_fuse=args.fuse, restart=True)
            bpr = record_bytes(bb, sp)
            bb.step(args.fuse, stream)                                       # pre-roll + warm-up in one launch
            bb.sync()
            barrier()
            t1 = time.perf_counter()
            for _ in range(k_timed):
                bb.step(args.fuse, stream)
            barrier()
            dt = max_over_ranks(time.perf_counter() - t1)
            t2 = time.perf_counter()
            sm = allgather_summary(bb, world, force=use_dist)
            barrier()
            ag_ms = (time.perf_counter() - t2) * 1e3
            turns_total = int(bb.turn)
            bb.close()
            steps_s = r * world * args.fuse * k_timed / dt
            alg = 2 * bpr * steps_s / 1e9                                    # whole job
            # every rank streams its share through single-turn launches too (C4: the configuration BASELINE names for 8 GPUs)
            stream_pt = None
            if key == "c4" and not args.no_unfused:
                barrier()
                stream_pt = streaming_point(sp, launches=64, first_room=shard_first_room(r, rank), preroll=256, parity=False)
                stream_pt["us_per_launch_sustained_max_over_ranks"] = max_over_ranks(stream_pt["us_per_launch_sustained"])
                stream_pt["frac_min_over_ranks"] = 2 * bpr * r / (stream_pt["us_per_launch_sustained_max_over_ranks"] * 1e-6) / 1e9 / HBM_PEAK_GBS
                barrier()
            other_workloads[key] = {
                "workload": " + ".join(f"{x[2]} {x[0]} rooms x {x[1]} players" for x in sp) + f" per GPU, x{world} GPUs, steady state",
                "rooms_total": r * world, "rooms_per_gpu": r, "turns_fused_per_launch": args.fuse, "launches_timed": k_timed,
                "turns_stepped": turns_total, "seed": SEED,
                "value": steps_s, "unit": "room-phase steps/s (whole job, wall, max over ranks)", "ms_per_launch": dt * 1e3 / k_timed,
                "bytes_per_room_record": bpr, "algorithmic_GBs": alg, "algorithmic_frac": alg / (HBM_PEAK_GBS * world),
                "roofline": {"bound": "hbm", "achieved": alg, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": alg / (HBM_PEAK_GBS * world),
                             "algorithmic": True, "bound_actual": "valu-issue", "traffic": None,
                             "issue": issue_block(committed_profile(key, args.fuse), r, args.fuse * k_timed, dt)},
                "hbm_streaming": stream_pt,
                "summary_allgather_ms": ag_ms,
                "checksum": sm["checksum"], "summary": {k: sm[k] for k in ("rooms", "finished", "village_wins", "wolf_wins", "games_recycled")}}

    if rank == 0:
        turns_timed = args.fuse * args.steps
        total_steps = rooms * world * turns_timed
        per_launch_units = rooms * (turns_timed / max(launches, 1))
        alg_bytes = 2 * bytes_per_room * per_launch_units
        avg_launch_s = kernel_ms * 1e-3 / max(launches, 1)
        achieved = alg_bytes / avg_launch_s / 1e9
        prof = committed_profile(args.workload if not args.rooms else f"ww8_{args.rooms}", args.fuse)
        state_rw = 2.0 * bytes_per_room * rooms
        traffic = prof.get("hbm_bytes_per_launch") if abs(prof.get("state_bytes_read_plus_written", -1) - state_rw) < 1 else None
        fused = args.fuse > 1
        issue = issue_block(prof, rooms, turns_timed, kernel_ms * 1e-3, (args.workload if not args.rooms else f"ww8_{args.rooms}") if fused else None)
        def phys(label, pt):
            """one row of roofline.physical: a single-turn launch that really moves the state, from this run's HIP events"""
            return {"shape": label, "rooms": pt["rooms"], "state_MiB": round(pt["resident_state_MiB"], 1),
                    "us_per_launch": round(pt["us_per_launch_sustained"], 2), "frac": round(pt["frac"], 4),
                    "hbm": not pt["fits_infinity_cache"], "hbm_floor_frac": round(pt["hbm_floor_frac"], 4),
                    "parity": (pt.get("parity") or {}).get("single_turn_equals_fused")}

        physical = []
        for label, pt in (beyond_l3 or {}).items():
            physical.append(phys(label, pt))
        if unfused:
            physical.append(phys(f"{rooms} {GAME} x{N_PLAYERS} ({args.workload})", unfused))
        for label, o in (other or {}).items():
            if o.get("hbm_streaming"):
                physical.append(phys(label, o["hbm_streaming"]))
        fused_rows = [{"shape": f"{rooms} {GAME} x{N_PLAYERS} ({args.workload})", "us_per_turn": round(kernel_ms * 1e3 / turns_timed, 3),
                       "steps_per_s": total_steps / elapsed, "valu_frac": issue.get("valu_frac"), "valu_priced_frac": issue.get("valu_priced_frac"), "issue_frac": issue.get("frac")}]
        for label, o in (other or {}).items():
            fused_rows.append({"shape": label, "us_per_turn": round(o["us_per_turn"], 3), "steps_per_s": o["value"],
                               "valu_frac": o["issue"].get("valu_frac"), "valu_priced_frac": o["issue"].get("valu_priced_frac"), "issue_frac": o["issue"].get("frac")})
        out = {
            "metric": "room-phase steps/sec", "value": total_steps / elapsed, "unit": "room-phase steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": args.workload + ": " + " + ".join(f"{r} {g} rooms x {n} players" for g, n, r in spec) +
                                   " per GPU, steady state (finished rooms recycled, untimed pre-roll), fixed policy, seed 0xC0FFEE",
                       "step": f"one fused launch = {args.fuse} turns of every room (record loaded once, stored once)",
                       "rooms_per_gpu": rooms, "n_players": [n for _, n, _ in spec], "turns_fused_per_launch": args.fuse,
                       "room_phase_steps_per_bench_step": rooms * world * args.fuse, "preroll_turns": preroll,
                       "bytes_per_room_record": bytes_per_room, "sharding": f"rooms x{world}, no data-path collective"},
            # the bulky per-shape blocks first, the contract's blocks last: a tail of the line still shows them
            "issue": issue,
            "hbm_streaming": unfused,
            "hbm_streaming_beyond_l3": beyond_l3,
            "from_init_64": from_init,
            "other_shapes": other,
            "other_workloads": other_workloads,
            "summary": {k: summary[k] for k in ("rooms", "finished", "village_wins", "wolf_wins", "games_recycled", "checksum")},
            "summary_allgather_ms": summary_ms,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak": achieved / HBM_COPY_GBS,
                         "traffic": traffic,
                         "traffic_source": (prof.get("_path", None) and f"{prof['_path']} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"),
                         "kernel": "ge_step_kernel", "avg_launch_us": avg_launch_s * 1e6, "launches": launches,
                         # `achieved` / `frac` are SURVEY 8(d)'s yardstick: ALGORITHMIC bytes = 2 x record x rooms x turns in the launch.
                         # Fused turns keep the state in registers: those bytes never reach HBM (`traffic`), the launch is bound by
                         # the vector pipe (`bound_actual`; `frac_of_actual_bound` = VALU busy fraction of the issue ceiling)
                         "algorithmic": True, "algorithmic_bytes_per_launch": alg_bytes,
                         "bound_actual": "valu-issue" if fused else "hbm",
                         "frac_of_actual_bound": issue.get("valu_frac") if fused else achieved / HBM_PEAK_GBS,
                         # ... and the same with every vector instruction at its measured price instead of a nominal 2 cycles (issue.valu_priced_frac)
                         "frac_of_priced_bound": issue.get("valu_priced_frac") if fused else None,
                         "hbm_frac_of_measured_traffic": (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         # the PHYSICAL figures of this run: single-turn launches (max_fuse = 1), every launch reads and writes every
                         # record; frac = state read + written / device time per launch (HIP events around a replayed hipGraph on the
                         # launch stream) / 8 TB/s; hbm: the resident state is larger than the 256 MiB Infinity Cache (>= hbm_floor_frac
                         # of 8 TB/s is then DRAM traffic); parity: 64 such launches == 64 fused turns (whole summary + checksum)
                         "physical": physical,
                         "fused": fused_rows},
        }
        if not args.no_cpu_baseline and args.workload == "c2":
            # rank 0 only.  N = 1: the contract's ~12 s sample; N > 1: a shorter one (the other ranks wait at the barrier below),
            # so that a multi-GPU record carries its own CPU row, plus one for the C4 shape
            out["cpu_baseline"] = cpu_baseline(spec, budget_s=12.0 if world == 1 else 5.0)
            if world > 1 and other_workloads:
                other_workloads["c4"]["cpu_baseline"] = cpu_baseline(WORKLOADS["c4"], budget_s=4.0, sample_rooms=1 << 18, single_thread=False)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    barrier()
    batch.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
