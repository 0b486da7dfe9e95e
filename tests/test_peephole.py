"""game_engine_amd/csrc/peephole.sed rewrites the compiler's three-input bitwise instructions to v_bitop3_b32 before the device code is
assembled (Makefile).  CPU checks: every rule's truth table IS the function of the instruction it replaces (v_bitop3's table is
indexed by the operands' bits: src0 = 0xF0, src1 = 0xCC, src2 = 0xAA), the rules keep operands and comments, and the assembly that is
shipped (make asm -> ge_step.s) has none of the old forms left."""
import os
import re
import subprocess

from conftest import ROOT

CSRC = os.path.join(ROOT, "game_engine_amd", "csrc")
A, B, C = 0xF0, 0xCC, 0xAA
SEMANTICS = {                                  # AMD "CDNA4 ISA": D = ...
    "v_or3_b32": (A | B | C),                  # S0 | S1 | S2
    "v_and_or_b32": ((A & B) | C),             # (S0 & S1) | S2
    "v_bfi_b32": ((A & B) | (~A & C)) & 0xFF,  # (S0 & S1) | (~S0 & S2)
    "v_xor3_b32": (A ^ B ^ C),                 # S0 ^ S1 ^ S2
}


def rules():
    out = {}
    with open(os.path.join(CSRC, "peephole.sed")) as f:
        for ln in f:
            m = re.match(r"s/.*\)(v_[a-z0-9_]+)\\\(.*bitop3:(0x[0-9a-f]+)", ln)
            if m:
                out[m.group(1)] = int(m.group(2), 16)
    return out


def test_every_rule_carries_the_truth_table_of_the_instruction_it_replaces():
    r = rules()
    assert set(r) == set(SEMANTICS)
    for op, table in r.items():
        assert table == SEMANTICS[op], (op, hex(table), hex(SEMANTICS[op]))


def test_rules_keep_operands_indentation_and_comments():
    text = ("\tv_or3_b32 v0, v0, v23, v22\n"
            "\tv_and_or_b32 v1, v2, 0xff, s3 ; a comment\n"
            "  v_bfi_b32 v60, v19, v20, v60\n"
            "\tv_xor3_b32 v1, v2, v3, v4\t\n"
            "\tv_or_b32_e32 v5, v0, v5\n"
            "\tv_lshl_or_b32 v23, v23, 8, v40\n"
            "; v_or3_b32 in a comment line stays\n")
    p = subprocess.run(["sed", "-f", os.path.join(CSRC, "peephole.sed")], input=text, stdout=subprocess.PIPE, text=True, check=True)
    assert p.stdout.splitlines() == [
        "\tv_bitop3_b32 v0, v0, v23, v22 bitop3:0xfe",
        "\tv_bitop3_b32 v1, v2, 0xff, s3 bitop3:0xea ; a comment",
        "  v_bitop3_b32 v60, v19, v20, v60 bitop3:0xca",
        "\tv_bitop3_b32 v1, v2, v3, v4 bitop3:0x96\t",
        "\tv_or_b32_e32 v5, v0, v5",
        "\tv_lshl_or_b32 v23, v23, 8, v40",
        "; v_or3_b32 in a comment line stays"]


def test_shipped_assembly_has_no_old_form_left():
    """(the file tools/asm_table.py's `make asm` leaves; test_asm_table.py has just rebuilt it when the suite runs in order, else build it)"""
    s_path = os.path.join(CSRC, "ge_step.s")
    if not os.path.exists(s_path):
        subprocess.run(["make", "-C", CSRC, "asm", "-s"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    with open(s_path) as f:
        txt = f.read()
    assert not re.search(r"^\s*v_(or3|and_or|bfi|xor3)_b32\s", txt, flags=re.M)
    assert len(re.findall(r"^\s*v_bitop3_b32\s", txt, flags=re.M)) > 2000
