#!/bin/bash
# A/B of issue priorities at the two ends of a single-turn launch's wavefront (profiles/r05_ab_k1_prio.txt).  The variants were builds of
# libge_step.so with -DGE_LOAD_PRIO=3 (s_setprio 3 from kernel entry until the record loads and the table fill are issued; that code was dropped after
# this measurement) and / or -DGE_STORE_PRIO=0 / 3 (kept: ge_kernels.inl), placed in game_engine_amd/ab/k_<name>.so and selected through GE_LIB_PATH.
for rep in 1 2 3; do for v in n l3 s3 l3s3 l3s1; do echo "== k_$v"; GE_LIB_PATH=$PWD/game_engine_amd/ab/k_$v.so timeout -k 10 200 python tools/k1_probe.py ww:8:1048576 ww:12:2097152 tt:4:1048576 ww:8:524288+tt:4:524288 ww:8:33554432 2>&1 | grep -v amdgpu; done; done
