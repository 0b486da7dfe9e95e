#!/bin/bash
# A/B of the instruction-price changes (DESIGN.md 4 "Attribution"): parity subset on each variant, then interleaved timings.  tools/price_ab.sh <tag> <variant.so> ...
set -u
TAG=$1; shift; mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ab_valu_price.txt; : > $OUT
for v in "$@"; do
  echo "# parity subset on $v" >> $OUT
  GE_LIB_PATH=$PWD/$v timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_generic_dsl.py -m gpu -q -x 2>&1 | tail -2 >> $OUT || exit 1
done
PROBE_FUSE="1024:4096" tools/abn.sh "ww:8:65536 ww:8:1048576 ww:12:2097152 tt:4:1048576 ww:8:524288+tt:4:524288" game_engine_amd/ab/product.so "$@" >> $OUT 2>&1
grep -c "us/turn" $OUT
