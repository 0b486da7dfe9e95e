# Device assembly between the compiler and the assembler (Makefile): every three-input bitwise instruction the compiler keeps in its
# older form becomes the v_bitop3_b32 of the same truth table (operands in the same order: the table is over src0 = 0xF0, src1 = 0xCC,
# src2 = 0xAA).  On gfx950 v_or3_b32 / v_and_or_b32 / v_bfi_b32 / v_xor3_b32 cost a SIMD 4.15 cycles per wavefront, v_bitop3_b32 2.45
# (tools/microbench/encoding_probe.hip, profiles/r05_encoding_probe.txt); the compiler prefers the former on purpose - for the
# readability of the disassembly - and has no switch for it.  Measured: profiles/r05_ab_peephole.txt, r05_ab_valu_price.txt.
s/^\([ \t]*\)v_or3_b32\([ \t][^;]*[^; \t]\)\([ \t]*\(;.*\)\{0,1\}\)$/\1v_bitop3_b32\2 bitop3:0xfe\3/
s/^\([ \t]*\)v_and_or_b32\([ \t][^;]*[^; \t]\)\([ \t]*\(;.*\)\{0,1\}\)$/\1v_bitop3_b32\2 bitop3:0xea\3/
s/^\([ \t]*\)v_bfi_b32\([ \t][^;]*[^; \t]\)\([ \t]*\(;.*\)\{0,1\}\)$/\1v_bitop3_b32\2 bitop3:0xca\3/
s/^\([ \t]*\)v_xor3_b32\([ \t][^;]*[^; \t]\)\([ \t]*\(;.*\)\{0,1\}\)$/\1v_bitop3_b32\2 bitop3:0x96\3/
