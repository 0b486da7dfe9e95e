#!/bin/bash
# A/B of two builds of libge_step.so on the same GPU box, interleaved (box-to-box variance is ~2 %):
#   tools/ab.sh game_engine_amd/ab/a.so game_engine_amd/ab/b.so ww:8:65536 [more shapes]
# Build the variants in the container first (they travel with the snapshot), e.g.
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGE_DPP_SCAN=0 -shared -o game_engine_amd/ab/a.so game_engine_amd/csrc/ge_step.hip game_engine_amd/csrc/ge_table.cpp
set -e
A=$1; B=$2; shift 2
cp game_engine_amd/libge_step.so /tmp/ge_keep.so
for rep in 1 2 3; do
  for v in "$A" "$B"; do
    cp "$v" game_engine_amd/libge_step.so
    echo "== $v"
    timeout -k 10 200 python tools/perf_probe.py "$@" | grep "fuse=64"
  done
done
cp /tmp/ge_keep.so game_engine_amd/libge_step.so
