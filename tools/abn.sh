#!/bin/bash
# A/B/n of several builds of libge_step.so on the same GPU box, interleaved (box-to-box variance is ~2 %):
#   tools/abn.sh "ww:8:65536 ww:8:1048576" game_engine_amd/ab/a.so game_engine_amd/ab/b.so ...
# Build the variants in the container first (they travel with the snapshot).  The loader is pointed at a variant through
# GE_LIB_PATH (game_engine_amd/_lib.py): the product library is never overwritten, whatever a variant does.
SHAPES=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    echo "== $v"
    GE_LIB_PATH="$PWD/$v" timeout -k 10 200 python tools/perf_probe.py $SHAPES || echo "(variant failed: rc=$?)"
  done
done
