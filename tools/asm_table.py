#!/usr/bin/env python3
"""Registers, spills, scratch, LDS and occupancy of every kernel of libge_step.so, from the compiler's own remarks
(`make -C game_engine_amd/csrc asm` = hipcc -S -Rpass-analysis=kernel-resource-usage): the source of DESIGN.md's register
table, so that the document cannot drift from the build.

    python tools/asm_table.py            run `make asm` and print the table (markdown)
    python tools/asm_table.py --json     the same as JSON (tools/design_tables.py reads this)
    python tools/asm_table.py --check    exit 1 if ANY kernel uses scratch or spills a vector register, or spills more scalar
                                         registers (to VGPR lanes: v_writelane / v_readlane, no memory) than the committed
                                         baseline tools/asm_baseline.json allows for it (--write-baseline records today's)
tests/test_asm_table.py runs --check on the CPU (hipcc -S needs no GPU), so the table of DESIGN.md and this claim stay pinned.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "game_engine_amd", "csrc")
KINDS = ["Werewolf x 8", "Werewolf x 12", "Two-Truths x 4", "Two-Truths x 8", "Two-Truths x 12"]
KEYS = {"Function Name": "name", "TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds_static"}


def describe(mangled: str) -> dict:
    """ge_step_kernel<KIND, LOWOCC, GENERIC, SINGLE, LD> / ge_step_kernel_mixed<LOWOCC, GENERIC, SINGLE> / the helper kernels"""
    m = re.search(r"ge_step_kernel_mixedILb([01])ELi(\d)ELb([01])E", mangled)
    if m:
        low, gen = m.group(1) == "1", int(m.group(2))
        return {"kernel": "ge_step_kernel_mixed", "layout": "mixed batch", "lowocc": low, "generic": gen, "single": m.group(3) == "1"}
    m = re.search(r"ge_step_kernelILi(\d)ELb([01])ELi(\d)ELb([01])ELi(\d)E", mangled)
    if m:
        return {"kernel": "ge_step_kernel", "layout": KINDS[int(m.group(1))], "lowocc": m.group(2) == "1", "generic": int(m.group(3)),
                "single": m.group(4) == "1", "ld": int(m.group(5))}
    m = re.search(r"N_1\d+(ge_[a-z_0-9]+?)E", mangled)
    return {"kernel": m.group(1) if m else mangled, "layout": "-", "lowocc": False, "generic": False, "single": False}


def collect(rebuild=True):
    cmd = ["make", "-C", CSRC, "asm"] + (["-B"] if rebuild else [])
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stdout)
        raise SystemExit("make asm failed")
    rows, cur = [], None
    for line in p.stdout.splitlines():
        m = re.search(r"remark: ([^:]+): (.*?) \[-Rpass-analysis", line)
        if not m or m.group(1).strip() not in KEYS:
            continue
        key, val = KEYS[m.group(1).strip()], m.group(2).strip()
        if key == "name":
            cur = {"name": val}
            cur.update(describe(val))
            rows.append(cur)
        elif cur is not None:
            cur[key] = int(val)
    return rows


def label(r):
    if r["kernel"] not in ("ge_step_kernel", "ge_step_kernel_mixed"):
        return r["kernel"]
    form = "single-turn" if r["single"] else "fused"
    occ = "lone-wavefront" if r["lowocc"] else "large-batch"
    shape = {0: "", 1: ", GENERIC", 2: ", GENERIC 1 x 1", 3: ", GENERIC 1 x 2"}[int(r["generic"])]
    return f"{r['layout']}, {occ}, {form}{shape}" + {0: "", 1: ", plain record loads", 2: ", streaming record loads"}[r.get("ld", 0)]


def markdown(rows):
    out = ["| build | VGPRs | SGPRs | SGPR spills | VGPR spills | scratch B/lane | wavefronts / SIMD |", "|---|---|---|---|---|---|---|"]
    order = sorted(rows, key=lambda r: (r["kernel"] not in ("ge_step_kernel", "ge_step_kernel_mixed"), r["generic"], r["single"], r["layout"], not r["lowocc"]))
    for r in order:
        out.append(f"| {label(r)} | {r.get('vgprs', 0)} | {r.get('sgprs', 0)} | {r.get('sgpr_spill', 0)} | {r.get('vgpr_spill', 0)} | "
                   f"{r.get('scratch', 0)} | {r.get('occupancy', 0)} |")
    return "\n".join(out)


def main():
    rows = collect(rebuild=True)
    if "--json" in sys.argv:
        print(json.dumps(rows, indent=1))
        return
    print(markdown(rows))
    base_path = os.path.join(ROOT, "tools", "asm_baseline.json")
    if "--write-baseline" in sys.argv:
        with open(base_path, "w") as f:
            json.dump({label(r): r.get("sgpr_spill", 0) for r in rows if r.get("sgpr_spill", 0)}, f, indent=1, sort_keys=True)
            f.write("\n")
    if "--check" in sys.argv:
        with open(base_path) as f:
            allowed = json.load(f)
        bad = [f"{label(r)}: scratch {r.get('scratch', 0)} B/lane, {r.get('vgpr_spill', 0)} VGPR spills" for r in rows
               if r.get("scratch", 0) or r.get("vgpr_spill", 0)]
        bad += [f"{label(r)}: {r.get('sgpr_spill', 0)} SGPR spills, baseline {allowed.get(label(r), 0)}" for r in rows
                if r.get("sgpr_spill", 0) > allowed.get(label(r), 0)]
        if bad:
            raise SystemExit("register check failed: " + "; ".join(bad))


if __name__ == "__main__":
    main()
