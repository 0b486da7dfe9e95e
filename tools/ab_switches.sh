#!/bin/bash
# Every compile-time switch that is left in the kernels (ge_device.h, ge_kernels.inl) is built once with a non-default value
# and run through the parity subset, so that no switch value rots untested.
#   tools/ab_switches.sh build        in the container: game_engine_amd/ab/sw_*.so (they travel with the gpurun snapshot)
#   tools/ab_switches.sh test <tag>   on the GPU box: parity subset + timing probe per variant -> gpurun_out/<tag>/switches.txt
set -u
VARIANTS=("deal8:-DGE_DEAL_PERIOD=8" "deal32:-DGE_DEAL_PERIOD=32" "ttq13:-DGE_TT_LOW_QUEUE_MIN=13" "ttq4:-DGE_TT_LOW_QUEUE_MIN=4"
          "ww12w5:-DGE_WW12_WAVES=5" "ww8w8:-DGE_WW8_WAVES=8" "genw7:-DGE_GENERIC_WAVES=7" "stamps:-DGE_STAMPS=1" "clock:-DGE_STAMPS=2"
          "lds_old:-DGE_RES_PACKED=0 -DGE_ROWS_SPLIT=0 -DGE_RES_ATOMIC64=0" "noprio:-DGE_QUEUE_PRIO=0 -DGE_RESOLVE_PRIO=0 -DGE_STORE_PRIO=0" "deal_in_shadow:-DGE_DEAL_EARLY=0")
cd "$(dirname "$0")/.."
case "${1:-}" in
build)
  mkdir -p game_engine_amd/ab
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}; flag=${v#*:}
    ( make -C game_engine_amd/csrc -s OUT=../ab/sw_$name.so B=build/sw_$name EXTRA="$flag" 2>&1 | grep -i "error" ; test -f game_engine_amd/ab/sw_$name.so ) \
      && echo "built sw_$name.so ($flag)" || echo "BUILD FAILED $name"
  done ;;
test)
  TAG=${2:-r05}; mkdir -p gpurun_out/$TAG; OUT=gpurun_out/$TAG/switches.txt; : > $OUT
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    echo "== sw_$name (${v#*:})" >> $OUT
    GE_LIB_PATH=$PWD/game_engine_amd/ab/sw_$name.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_generic_dsl.py -q -x -m gpu -k "batch_equals_oracle or restart or variant_batches" 2>&1 | tail -1 >> $OUT
    GE_LIB_PATH=$PWD/game_engine_amd/ab/sw_$name.so timeout -k 10 200 python tools/perf_probe.py ww:8:65536 ww:8:1048576 ww:12:2097152 tt:4:1048576 2>&1 | grep "fuse=64" >> $OUT
  done
  cat $OUT ;;
*) echo "usage: $0 build | test <tag>"; exit 2 ;;
esac
