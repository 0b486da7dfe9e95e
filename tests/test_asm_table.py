"""Registers, spills and scratch of EVERY kernel of libge_step.so, from the compiler's own remarks (CPU: hipcc -S needs no GPU):
`tools/asm_table.py --check` fails on scratch, on a spilled vector register, or on more scalar spills (to VGPR lanes) than
tools/asm_baseline.json records for that build - and the register table of DESIGN.md is the one this build produces."""
import os
import re
import subprocess
import sys

from conftest import ROOT


def test_no_kernel_uses_scratch_and_the_design_table_is_this_builds():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_table.py"), "--check"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    table = [ln for ln in p.stdout.splitlines() if ln.startswith("|")]
    assert len(table) > 50                                       # 10 layouts x {fused, single-turn} x {shipped, GENERIC} + mixed + helpers
    with open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8") as f:
        m = re.search(r"<!-- gen:asm_table -->\n(.*?)<!-- /gen -->", f.read(), re.S)
    assert m and [ln for ln in m.group(1).splitlines() if ln.startswith("|")] == table, "DESIGN.md's register table is stale: tools/design_tables.py --write"


def test_priced_instruction_mix_is_of_this_device_code():
    """profiles/valu_mix.json (tools/valu_mix.py --write) carries the hash of the device code it priced: bench.py quotes it as
    `valu_priced_frac` only then, so a kernel change without re-pricing shows up here and not as a silently missing figure."""
    import json
    from game_engine_amd._lib import kernel_source_hash
    with open(os.path.join(ROOT, "profiles", "valu_mix.json")) as f:
        d = json.load(f)
    assert d["kernel_src_sha256"] == kernel_source_hash(), "run `make -C game_engine_amd/csrc asm && python tools/valu_mix.py --write`"
    assert set(d["mean_price_cycles"]) >= {"c2", "ww8_1048576", "c4", "c3", "c5"}
    assert all(2.0 < v < 4.3 for v in d["mean_price_cycles"].values())
