#!/usr/bin/env python3
"""The vector instructions of the step kernels, priced by what each KIND really costs a SIMD on gfx950 (tools/microbench/encoding_probe.hip,
profiles/r05_encoding_probe.txt; 8 wavefronts per SIMD, independent chains, cycles per wave-instruction):
    2.3   simple two-input integer ops on VGPRs / constants / literals: v_add / sub / and / or / xor / shifts / not (VOP2 or _e64 alike), v_mov 2.0
    2.45  v_bitop3_b32 (any three-input bitwise function)
    2.0   v_cndmask_b32 reading vcc (measured as 6.1 for the v_cmp + v_cndmask pair); 4.2 with an SGPR-pair mask (_e64)
    4.1   v_cmp_* (to vcc or to an SGPR pair)
    4.15  everything else: three-operand fused ops (v_and_or, v_or3, v_lshl_or, v_lshl_add, v_add3, v_bfi, v_perm, v_bfe, v_alignbit), multiplies
          (v_mul_lo / hi, v_mad_u32_u24, v_mul_u32_u24), v_bcnt, v_ffbl / ffbh, 64-bit shifts, DPP, and ANY simple op that takes an SGPR operand
Reads game_engine_amd/csrc/ge_step.s (`make -C game_engine_amd/csrc asm`) and prints, per fused kernel, the STATIC mix and its mean price.  With the
SQ counters' vector instructions per wave-turn (profiles/pmc_<shape>.json) that mean price gives the vector pipe's busy cycles per wave-turn, to set
against the SIMD cycles a wave-turn takes (profiles/<tag>_<shape>_attrib_counters.json: 4 x SQ_BUSY_CU_CYCLES).   python tools/valu_mix.py [tag]"""
import collections, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMPLE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_not_b32",
          "v_add_co_u32", "v_max_u32", "v_min_u32", "v_xnor_b32"}


def price(op, args):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    srcs = args.split(",")[1:]
    sgpr = any(re.search(r"\bs\d+\b|\bs\[\d+|\bvcc|\bexec|\bm0", s) for s in srcs)
    dpp = "dpp" in op or "row_" in args or "quad_perm" in args
    if base.startswith("v_cmp"):
        return 4.1, "compare"
    if base == "v_cndmask_b32":
        return (4.2, "select, SGPR-pair mask") if op.endswith("e64") else (2.0, "select, vcc")
    if base == "v_bitop3_b32":
        return 2.45, "bitop3"
    if base == "v_mov_b32" and not dpp:
        return (4.15, "simple op with an SGPR operand") if sgpr else (2.04, "simple")
    if base in SIMPLE and not dpp:
        return (4.15, "simple op with an SGPR operand") if sgpr else (2.3, "simple")
    if base in ("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_u32_u24", "v_mad_u64_u32"):
        return 4.2, "multiply"
    if base in ("v_or3_b32", "v_and_or_b32", "v_bfi_b32", "v_xor3_b32"):
        return 4.15, "three-input bitwise NOT as bitop3"
    return 4.15, "other half-rate (fused 3-operand, bfe, perm, bcnt, ffbl, 64-bit, DPP ...)"


KEYS = {"ww8_1048576": ("1 048 576 Werewolf × 8", "ILi0ELb0ELi0ELb0ELi0E"), "c4": ("C4 share: 2 097 152 Werewolf × 12", "ILi1ELb0ELi0ELb0ELi0E"),
        "c3": ("C3: 1 048 576 Two-Truths × 4", "ILi2ELb0ELi0ELb0ELi0E"),
        "c5": ("C5 share: 524 288 Werewolf × 8 + 524 288 Two-Truths × 4, one launch", "_mixedILb0ELi0ELb0E"),
        "c2": ("C2: 65 536 Werewolf × 8, a lone wavefront per SIMD", "ILi0ELb1ELi0ELb0ELi0E")}
LONE = 5.2          # a lone wavefront: cycles per vector instruction of any kind (the probe's 1-wavefront column)
OTHER = "other half-rate (fused 3-operand, bfe, perm, bcnt, ffbl, 64-bit, DPP ...)"


def mix(txt, mangled):
    """(instructions by kind, priced cycles) of the vector instructions of one kernel of ge_step.s"""
    i = txt.index("ge_step_kernel" + mangled + "E")
    body = txt[txt.rfind("\n", 0, i):txt.index("s_endpgm", i)]
    n = collections.Counter(); cyc = 0.0
    for ln in body.splitlines():
        m = re.match(r"\s*(v_[a-z0-9_]+)\s+(.*)", ln)
        if m:
            c, kind = price(m.group(1), m.group(2).split(";")[0])
            n[kind] += 1; cyc += c
    return n, cyc


def table(tag="r05"):
    txt = open(os.path.join(ROOT, "game_engine_amd", "csrc", "ge_step.s")).read()
    rows = ["| fused kernel | vector instructions in the binary | of them: simple / `v_bitop3` / selects on vcc, on an SGPR pair / compares / multiplies / other half-rate / simple but with an SGPR operand | mean price, cycles | × vector instructions per wave-turn = pipe cycles | SIMD cycles per wave-turn (measured) | vector pipe busy at these prices |",
            "|---|---|---|---|---|---|---|"]
    for key, (label, mangled) in KEYS.items():
        n, cyc = mix(txt, mangled)
        tot = sum(n.values()); mean = cyc / tot
        try:
            valu = json.load(open(os.path.join(ROOT, "profiles", f"pmc_{key}.json")))["instructions_per_wave_turn"]["valu"]
            simd = 4.0 * json.load(open(os.path.join(ROOT, "profiles", f"{tag}_{key}_attrib_counters.json")))["SQ_BUSY_CU_CYCLES"]["per_wave_turn"]
            if key == "c2":
                tail = f"{valu:.0f} × {LONE} (a lone wavefront: any kind) = {valu * LONE:.0f} | {simd:.0f} | {100 * valu * LONE / simd:.0f} % |"
            else:
                tail = f"{valu:.0f} × {mean:.2f} = {valu * mean:.0f} | {simd:.0f} | {100 * valu * mean / simd:.0f} % |"
        except (OSError, KeyError):
            tail = "- | - | - |"
        g = lambda k: n.get(k, 0)
        rows.append(f"| {label} | {tot} | {g('simple')} / {g('bitop3')} / {g('select, vcc')}, {g('select, SGPR-pair mask')} / {g('compare')} / {g('multiply')} / "
                    f"{g(OTHER) + g('three-input bitwise NOT as bitop3')} / {g('simple op with an SGPR operand')} | {mean:.2f} | {tail}")
    return rows


def write():
    """profiles/valu_mix.json: the mean price per fused kernel, tied to the device code by its hash (bench.py quotes it as `valu_priced_frac`)"""
    sys.path.insert(0, ROOT)
    from game_engine_amd._lib import kernel_source_hash
    txt = open(os.path.join(ROOT, "game_engine_amd", "csrc", "ge_step.s")).read()
    out = {"kernel_src_sha256": kernel_source_hash(), "prices_from": "profiles/r05_encoding_probe.txt (tools/microbench/encoding_probe.hip)",
           "what": "mean cycles a SIMD spends per vector instruction of the kernel's binary (static mix) at >= 4 wavefronts per SIMD; "
                   "a lone wavefront issues one vector instruction of any kind per lone_wavefront_cycles",
           "lone_wavefront_cycles": LONE, "mean_price_cycles": {}, "vector_instructions_in_binary": {}}
    for key, (_, mangled) in KEYS.items():
        n, cyc = mix(txt, mangled)
        out["mean_price_cycles"][key] = round(cyc / sum(n.values()), 3)
        out["vector_instructions_in_binary"][key] = sum(n.values())
    with open(os.path.join(ROOT, "profiles", "valu_mix.json"), "w") as f:
        json.dump(out, f, indent=1); f.write("\n")
    print(json.dumps(out["mean_price_cycles"]))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--write" in sys.argv:
        write()
    else:
        print("\n".join(table(args[0] if args else "r05")))
