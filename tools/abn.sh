#!/bin/bash
# A/B/n of several builds of libge_step.so on the same GPU box, interleaved (box-to-box variance is ~2 %):
#   tools/abn.sh "ww:8:65536 ww:8:1048576" game_engine_amd/ab/a.so game_engine_amd/ab/b.so ...
# Build the variants in the container first (they travel with the snapshot).
set -e
SHAPES=$1; shift
cp game_engine_amd/libge_step.so /tmp/ge_keep.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp "$v" game_engine_amd/libge_step.so
    echo "== $v"
    timeout -k 10 200 python tools/perf_probe.py $SHAPES | grep "fuse=64"
  done
done
cp /tmp/ge_keep.so game_engine_amd/libge_step.so
