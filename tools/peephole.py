import re, sys
TT = {"v_or3_b32": "0xfe", "v_and_or_b32": "0xea", "v_bfi_b32": "0xca", "v_xor3_b32": "0x96"}
n = 0
out = []
for ln in open(sys.argv[1]):
    m = re.match(r"(\s*)(v_or3_b32|v_and_or_b32|v_bfi_b32|v_xor3_b32)(\s+)([^;\n]*?)(\s*(;.*)?)\n", ln)
    if m:
        ln = f"{m.group(1)}v_bitop3_b32{m.group(3)}{m.group(4)} bitop3:{TT[m.group(2)]}{m.group(5)}\n"; n += 1
    out.append(ln)
open(sys.argv[2], "w").writelines(out)
print("rewrote", n)
