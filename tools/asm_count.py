#!/usr/bin/env python3
"""Static instruction counts of the step kernels in game_engine_amd/csrc/ge_step.s (make asm):
VALU / SALU / LDS per kernel and for the fused-turn loop (outermost loop body)."""
import re, sys
src = open(sys.argv[1] if len(sys.argv) > 1 else "game_engine_amd/csrc/ge_step.s").read().split("\n")
starts = [(i, l) for i, l in enumerate(src) if re.match(r"^_ZN.*ge_step_kernel\S*:", l)]
for i, name in starts:
    end = next(j for j in range(i, len(src)) if "s_endpgm" in src[j])
    body = src[i:end]
    loop0 = next((k for k, l in enumerate(body) if "Loop Header: Depth=1" in l and "Inner" not in l), None)
    def cnt(seg):
        return tuple(sum(1 for x in seg if re.match(r"\s+" + p, x)) for p in ("v_", "s_", "ds_", "s_cbranch"))
    tot = cnt(body)
    # the loop body = from its header to the last line that mentions it as parent/header
    last = max(k for k, l in enumerate(body) if "Header=BB" in l or "Loop Header" in l or "Parent Loop" in l)
    lp = cnt(body[loop0:last + 40]) if loop0 is not None else (0, 0, 0, 0)
    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:40]
    print(f"{short:42s} total valu {tot[0]:4d} salu {tot[1]:4d} lds {tot[2]:3d} br {tot[3]:3d} | loop~ valu {lp[0]:4d} salu {lp[1]:4d} lds {lp[2]:3d} br {lp[3]:3d}")
