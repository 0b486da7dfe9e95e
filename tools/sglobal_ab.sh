#!/bin/bash
# A/B of GE_SINGLE_GLOBAL (which single-turn builds read the tables from global memory instead of filling LDS behind a barrier; ge_device.h):
#   make -C game_engine_amd/csrc OUT=../ab/sg<mask>.so B=build/sg<mask> EXTRA=-DGE_SINGLE_GLOBAL=<mask>;  tools/sglobal_ab.sh <tag> "<shapes>" product sg13 sg9 ...
set -u
TAG=$1; SHAPES=$2; shift 2
mkdir -p gpurun_out; OUT=gpurun_out/${TAG}_ab_single_global.txt; : > $OUT
for v in "$@"; do
  [ $v = product ] && continue
  echo "# parity (single-turn tests) on $v.so" >> $OUT
  GE_LIB_PATH=$PWD/game_engine_amd/ab/$v.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_generic_dsl.py tests/test_gpu_abi_sequences.py -m gpu -q -x -k "single or k1 or beyond or graph or fuzz" 2>&1 | tail -3 >> $OUT || exit 1
done
for rep in 1 2 3; do for v in "$@"; do echo "== $v" >> $OUT
  GE_LIB_PATH=$PWD/game_engine_amd/ab/$v.so timeout -k 10 200 python tools/k1_probe.py $SHAPES 2>&1 | grep -v amdgpu >> $OUT
done; done
cat $OUT
