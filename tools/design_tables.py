#!/usr/bin/env python3
"""The measurement tables of DESIGN.md §4 / BASELINE.md §5 from the committed records: profiles/pmc_<shape>{,_k1}.json
(tools/pmc_summary.py) and the driver-flags bench line profiles/<tag>_bench_n1_driver_flags.json.
usage: design_tables.py [tag] [--write]    (--write: rewrite the marked blocks of DESIGN.md / BASELINE.md in place -
`<!-- gen:NAME -->` ... `<!-- /gen -->` blocks and `<!--g:NAME-->value<!--/g-->` spans - instead of printing)"""
import io, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = args[0] if args else "r05"
_blocks, _real_print, _cur = {}, print, [None]
def begin(name): _cur[0] = name; _blocks[name] = []
def print(*a):                                                   # rows go to the current block
    if a: _blocks[_cur[0]].append(" ".join(str(x) for x in a))
P = lambda k: json.load(open(os.path.join(root, "profiles", f"pmc_{k}.json")))
bench = json.loads(open(os.path.join(root, "profiles", f"{tag}_bench_n1_driver_flags.json")).read().strip().splitlines()[-1])
other = bench["other_shapes"]
labels = {"c2": "C2: 65 536 Werewolf × 8 (1 wave/SIMD, ceiling 6.1·10¹¹)", "ww8_1048576": "1 048 576 Werewolf × 8",
          "c4": "C4 share: 2 097 152 Werewolf × 12", "c3": "C3: 1 048 576 Two-Truths × 4",
          "c5": "C5 share: 524 288 Werewolf × 8 + 524 288 Two-Truths × 4, one launch"}
okey = {"ww8_1048576": "1048576 Werewolf x8", "c4": "2097152 Werewolf x12 (one GPU's share of C4)", "c3": "1048576 Two-Truths x4 (C3)",
        "c5": "524288 Werewolf x8 + 524288 Two-Truths x4 (one GPU's share of C5)"}
rooms = {"c2": 65536, "ww8_1048576": 1 << 20, "c4": 1 << 21, "c3": 1 << 20, "c5": 1 << 20}
SHAPES = ("c2", "ww8_1048576", "c4", "c3", "c5") if okey["c5"] in other else ("c2", "ww8_1048576", "c4", "c3")

begin("fused_table")
print("| shape (fused, 1 024 turns/launch) | VALU + SALU + LDS per wave-turn | µs per turn | room-phase steps/s | issue frac (all / VALU only) | `SQ_WAIT_ANY` of wave cycles |\n|---|---|---|---|---|---|")
for k in SHAPES:
    p = P(k); i = p["instructions_per_wave_turn"]
    us = bench["roofline"]["avg_launch_us"] / 1024 if k == "c2" else other[okey[k]]["us_per_turn"]
    waves = rooms[k] // 64
    ceil = 1024 * 2.4e9 / (4.0 if waves / 1024 < 2 else 2.0)
    wt = waves / (us * 1e-6)
    tot = i["valu"] + i["salu"] + i["lds"]
    print(f"| {labels[k]} | {i['valu']:.0f} + {i['salu']:.0f} + {i['lds']:.0f} | {us:.3f} | {rooms[k] / (us * 1e-6):.3g} | {tot * wt / ceil:.2f} / {i['valu'] * wt / ceil:.2f} | {100 * p['wait_any_frac']:.0f} % |")
begin("k1_table")
print("| shape (single-turn launches; state <= 256 MiB: a memory-side rate, the Infinity Cache may serve) | VALU + SALU + LDS per wave-turn | measured memory-side bytes per launch (state read + written) | kernel-trace average (sustained) | % of 8 TB/s by kernel-trace | bench line: sustained / per-launch events | `SQ_WAIT_ANY` |\n|---|---|---|---|---|---|---|")
for k in SHAPES:
    p = P(k + "_k1"); i = p["instructions_per_wave_turn"]; kt = p["kernel_trace"]
    hs = bench["hbm_streaming"] if k == "c2" else other[okey[k]]["hbm_streaming"]
    st = p["state_bytes_read_plus_written"]
    print(f"| {labels[k].split(' (1 wave')[0]} | {i['valu']:.0f} + {i['salu']:.0f} + {i['lds']:.0f} | {p['hbm_bytes_per_launch'] / 1e6:.1f} MB ({st / 1e6:.1f}) | {kt['average_ns'] / 1e3:.2f} µs ({kt['calls']} launches) | **{100 * st / kt['average_ns'] / 8e3:.1f}** | {100 * hs['frac']:.1f} / {100 * hs['frac_kernel']:.1f} | {100 * p['wait_any_frac']:.0f} % |")
begin("beyond_l3_table")
# single-turn launches over a resident state larger than the Infinity Cache: the HBM figure that is provably HBM
print("| shape (single-turn launches, resident state > 256 MiB) | resident state | VALU + SALU + LDS per wave-turn | measured HBM bytes per launch (state read + written) | kernel-trace average | **% of 8 TB/s by kernel-trace** (% of the 6.29 TB/s copy rate) | bench line (HIP events around a graph replay) | 64 single-turn launches == 64 fused turns |\n|---|---|---|---|---|---|---|---|")
for label, key in (("33554432 Werewolf x8 (1 GiB of records)", "ww8_33554432"), ("16777216 Werewolf x12 (the WHOLE of C4 on one GPU, 640 MiB)", "c4_whole"), ("33554432 Two-Truths x4 (768 MiB)", "tt4_33554432")):
    p = P(key + "_k1"); i = p["instructions_per_wave_turn"]; kt = p["kernel_trace"]; st = p["state_bytes_read_plus_written"]
    b = bench["hbm_streaming_beyond_l3"][label]
    pct = 100 * st / kt["average_ns"] / 8e3
    print(f"| {label.replace(' x', ' × ')} | {b['resident_state_MiB']:.0f} MiB | {i['valu']:.0f} + {i['salu']:.0f} + {i['lds']:.0f} | {p['hbm_bytes_per_launch'] / 1e6:.1f} MB ({st / 1e6:.1f}) | {kt['average_ns'] / 1e3:.1f} µs ({kt['calls']} launches) | **{pct:.1f}** ({pct * 8 / 6.29:.0f}) | {100 * b['frac']:.1f} ({b['us_per_launch_sustained']:.1f} µs) | {'yes' if b['parity']['single_turn_equals_fused'] else 'NO'} |")
begin("attrib_table")
# where the fused launches' time goes: counters of tools/attrib_profile.sh (profiles/<tag>_<shape>_attrib_counters.json), per wave-turn
def A(k): return json.load(open(os.path.join(root, "profiles", f"{tag}_{k}_attrib_counters.json")))
print("| shape (fused) | SIMD cycles per wave-turn | vector instructions x the nominal 2 cycles | active lanes per VALU instruction | LDS array busy (of it bank-conflict replays) | a wavefront's residency: issuing / issue-stalled (of it on the LDS) / parked at a wait | mean resident wavefronts per SIMD | instruction fetches, I-cache misses |\n|---|---|---|---|---|---|---|---|")
for k in SHAPES:
    try: a = A(k)
    except OSError: continue
    g = lambda n: a[n]["per_wave_turn"]
    simd = 4.0 * g("SQ_BUSY_CU_CYCLES")                      # CU-busy cycles per wave-turn x the CU's 4 SIMDs working side by side
    wave = 4.0 * g("SQ_WAVE_CYCLES")                          # quad-cycles -> cycles
    print(f"| {labels[k].split(' (1 wave')[0]} | {simd:.0f} | {100 * 2 * g('SQ_INSTS_VALU') / simd:.0f} % | {g('SQ_THREAD_CYCLES_VALU') / g('SQ_INSTS_VALU'):.0f} of 64 | "
          f"{100 * g('SQ_LDS_IDX_ACTIVE') / g('SQ_BUSY_CU_CYCLES'):.0f} % ({100 * g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.0f} %) | "
          f"{100 * g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES'):.0f} % / {100 * g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.0f} % ({100 * g('SQ_WAIT_INST_LDS') / g('SQ_WAVE_CYCLES'):.0f} %) / {100 * g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.0f} % | "
          f"{wave / simd:.1f} | {g('SQ_IFETCH'):.0f}, {g('SQC_ICACHE_MISSES'):.2f} |")
begin("asm_table")
# registers, spills, scratch, occupancy of every kernel, from the compiler's remarks (tools/asm_table.py = make asm)
sys.path.insert(0, os.path.join(root, "tools"))
import asm_table as _asm
for line in _asm.markdown(_asm.collect(rebuild=True)).splitlines(): print(line)
begin("valu_mix")
# the vector instructions priced by what each kind costs a SIMD (tools/microbench/encoding_probe.hip): tools/valu_mix.py over the ge_step.s just rebuilt
import valu_mix as _mix
for line in _mix.table(tag): print(line)
begin("result_table")
print("| shape | fused (1 024 turns/launch): steps/s | alg. GB/s (% of 8 TB/s: a yardstick, not traffic) | single-turn launches: memory-side % of 8 TB/s (sustained; state fits the Infinity Cache) | CPU: oracle, steps/s (cores) |\n|---|---|---|---|---|")
print(f"| C2: 65 536 Werewolf × 8 — the `bench.py` line | **{bench['value']:.3g}** (wall) | {bench['roofline']['achieved']:.0f} ({100 * bench['roofline']['frac']:.1f}) | {100 * bench['hbm_streaming']['frac']:.1f} (launch-bound: {bench['hbm_streaming']['us_per_launch_sustained']:.1f} µs per launch) | {bench['cpu_baseline']['value']:.3g} ({bench['cpu_baseline']['cores']}); one thread {bench['cpu_baseline']['single_thread_value']:.3g} |")
for k in SHAPES[1:]:
    v = other[okey[k]]
    print(f"| {labels[k]} | {v['value']:.3g} | {v['algorithmic_GBs']:.0f} ({100 * v['algorithmic_frac']:.0f}) | **{100 * v['hbm_streaming']['frac']:.1f}** ({v['hbm_streaming']['us_per_launch_sustained']:.2f} µs per launch) | {v['cpu_baseline']['value']:.3g} ({v['cpu_baseline']['cores']}) |")
begin("baseline_rows")
# BASELINE.md's round table: one row per shape
blabels = {"c2": "C2 65 536 Werewolf×8", "ww8_1048576": "1 048 576 Werewolf×8", "c4": "C4 share 2 097 152 Werewolf×12", "c3": "C3 1 048 576 Two-Truths×4",
           "c5": "C5 share 524 288 Werewolf×8 + 524 288 Two-Truths×4"}
for k in SHAPES:
    p = P(k); i = p["instructions_per_wave_turn"]; q = P(k + "_k1")
    us = bench["roofline"]["avg_launch_us"] / 1024 if k == "c2" else other[okey[k]]["us_per_turn"]
    waves = rooms[k] // 64
    ceil = 1024 * 2.4e9 / (4.0 if waves / 1024 < 2 else 2.0)
    wt = waves / (us * 1e-6)
    issue = f"{(i['valu'] + i['salu'] + i['lds']) * wt / ceil:.2f} / {i['valu'] * wt / ceil:.2f}"
    kt = 100 * q["state_bytes_read_plus_written"] / q["kernel_trace"]["average_ns"] / 8e3
    fused_mb = f"{p['hbm_bytes_per_launch'] / 1e6:.2f} MB" if k == "c2" else f"{p['hbm_bytes_per_launch'] / 1e6:.1f} MB"
    if k == "c2":
        r = bench["roofline"]; hs = bench["hbm_streaming"]; cb = bench["cpu_baseline"]
        print(f"| {blabels[k]} | **{bench['value']:.3g}** (`roofline.frac` {r['frac']:.4f}; {r['avg_launch_us']:.0f} µs per 1 024-turn launch) | {r['achieved']:.0f} ({100 * r['frac']:.1f}) | {issue} | "
              f"{kt:.1f} / {100 * hs['frac']:.1f} (launch-bound: {hs['us_per_launch_sustained']:.1f} µs per launch) | {fused_mb} | {cb['value']:.3g} (one thread {cb['single_thread_value']:.3g}) |")
    else:
        v = other[okey[k]]
        note = " (= the state: no scratch any more)" if k == "c4" else ""
        print(f"| {blabels[k]} | {v['value']:.3g} | {v['algorithmic_GBs']:.0f} ({100 * v['algorithmic_frac']:.0f}) | {issue} | **{kt:.1f}** / {100 * v['hbm_streaming']['frac']:.1f} | {fused_mb}{note} | {v['cpu_baseline']['value']:.3g} |")

# single values quoted in the prose
_i = P("c2")["instructions_per_wave_turn"]
_vals = {
    "ww8": "%.1f" % (other[okey["ww8_1048576"]]["value"] / 1e10), "c4": "%.1f" % (other[okey["c4"]]["value"] / 1e10),
    "c2": "%.2f" % (bench["value"] / 1e10), "c4fused": "%.3g" % other[okey["c4"]]["value"],
    "c2cpi": "%.1f" % (bench["roofline"]["avg_launch_us"] / 1024 * 1e-6 * 2.4e9 / (_i["valu"] + _i["salu"] + _i["lds"])),
    **{"k1_" + k.split("_")[0]: "%.1f" % (100 * P(k + "_k1")["state_bytes_read_plus_written"] / P(k + "_k1")["kernel_trace"]["average_ns"] / 8e3)
       for k in ("ww8_1048576", "c4", "c3")},
    "ww8f": "%.2f" % (other[okey["ww8_1048576"]]["value"] / 1e11), "c3f": "%.2f" % (other[okey["c3"]]["value"] / 1e11),
    "k1pct": " / ".join("%.1f" % (100 * P(k + "_k1")["state_bytes_read_plus_written"] / P(k + "_k1")["kernel_trace"]["average_ns"] / 8e3)
                        for k in ("ww8_1048576", "c4", "c3")) + " %",
}
for _lab, _key in (("33554432 Werewolf x8 (1 GiB of records)", "bl3_ww8"), ("16777216 Werewolf x12 (the WHOLE of C4 on one GPU, 640 MiB)", "bl3_c4"), ("33554432 Two-Truths x4 (768 MiB)", "bl3_tt4")):
    _q = P({"bl3_ww8": "ww8_33554432", "bl3_c4": "c4_whole", "bl3_tt4": "tt4_33554432"}[_key] + "_k1")
    _vals[_key] = "%.1f" % (100 * _q["state_bytes_read_plus_written"] / _q["kernel_trace"]["average_ns"] / 8e3)
_n2 = os.path.join(root, "profiles", f"{tag}_bench_n2_gloo_rehearsal.json")
if os.path.exists(_n2):                                           # the 2-rank rehearsal's whole-job figures
    n2 = json.loads(open(_n2).read().strip().splitlines()[-1]); ow = n2.get("other_workloads") or {}
    _vals["n2_c2"] = "%.3g" % n2["value"]
    for k in ("c4", "c5"):
        if k in ow: _vals["n2_" + k] = "%.3g" % ow[k]["value"]
    if "c4" in ow: _vals["n2_ag"] = "%.2f" % ow["c4"]["summary_allgather_ms"]
if "--write" in sys.argv:
    for doc in ("DESIGN.md", "BASELINE.md", "README.md", os.path.join("profiles", "README.md")):
        path = os.path.join(root, doc); s = open(path, encoding="utf-8").read(); before = s
        for name, rows in _blocks.items():
            s = re.sub(r"(<!-- gen:%s -->\n).*?(<!-- /gen -->)" % name, lambda m: m.group(1) + "\n".join(rows) + "\n" + m.group(2), s, flags=re.S)   # (also an empty block)
        for name, v in _vals.items():
            s = re.sub(r"(<!--g:%s-->).*?(<!--/g-->)" % name, lambda m: m.group(1) + v + m.group(2), s)
        if s != before:
            open(path, "w", encoding="utf-8").write(s); _real_print("rewrote", doc)
else:
    _real_print("\n\n".join("\n".join(rows) for rows in _blocks.values()))
    _real_print("\n" + json.dumps(_vals, ensure_ascii=False))
