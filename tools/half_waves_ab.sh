#!/bin/bash
# Half-filled lone wavefronts (32 rooms per wavefront for single-game batches of <= 32 768 rooms; csrc/ge_step.hip launch_geometry): GE_HALF_WAVES=0 (off),
# default (on up to 32 768 rooms), 1 (forced on up to 65 536 rooms), fused (1 024 turns per launch) and single-turn launches, three interleaved repetitions
SHAPES=${1:-"ww:8:4096 ww:8:16384 ww:8:32768 ww:8:49152 ww:8:65536 ww:12:32768 ww:12:65536 tt:4:32768 tt:8:32768 tt:12:32768"}
for rep in 1 2 3; do
  for mode in 0 default 1; do
    echo "== half=$mode"
    if [ $mode = default ]; then PROBE_FUSE=1024:4096,1:256 python tools/perf_probe.py $SHAPES; else GE_HALF_WAVES=$mode PROBE_FUSE=1024:4096,1:256 python tools/perf_probe.py $SHAPES; fi
  done
done
