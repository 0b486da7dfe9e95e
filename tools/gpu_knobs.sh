#!/bin/bash
# The whole -m gpu suite with each kernel build / launch shape forced over every batch size (no build may depend on the
# batch size it is normally chosen for).  tools/gpu_knobs.sh <tag>
set -u
TAG=${1:-r05}; mkdir -p gpurun_out/$TAG
run() { name=$1; shift; env "$@" timeout -k 10 280 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_knobs.py > gpurun_out/$TAG/knob_$name.log 2>&1; echo "$name rc=$? $(tail -1 gpurun_out/$TAG/knob_$name.log)"; }
run lowocc_never GE_LOWOCC_ROOMS=0
run lowocc_always GE_LOWOCC_ROOMS=99999999
run block256 GE_BLOCK_THREADS=256
run block64_nograph GE_BLOCK_THREADS=64 GE_NO_GRAPH=1
run single_block512 GE_SINGLE_BLOCK=512 GE_BLOCK_THREADS=256 GE_LOWOCC_ROOMS=0
run single_block1024 GE_SINGLE_BLOCK=1024 GE_BLOCK_THREADS=256 GE_LOWOCC_ROOMS=0
run no_generic_shapes GE_NO_GENERIC_SHAPES=1
run half_waves_on GE_HALF_WAVES=1
run half_waves_off GE_HALF_WAVES=0
run nt_loads_on GE_NT_LOADS=1
run nt_loads_off GE_NT_LOADS=0
