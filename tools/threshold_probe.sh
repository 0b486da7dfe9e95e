#!/bin/bash
# which build (lone-wavefront / large-batch) is faster at a batch size: tools/threshold_probe.sh [rooms ...]
for rooms in ${@:-57344 65536 69632 73728 81920}; do
  for low in 0 99999999; do
    echo "rooms=$rooms GE_LOWOCC_ROOMS=$low: $(GE_LOWOCC_ROOMS=$low python tools/perf_probe.py ww:8:$rooms | grep fuse=64 | awk '{print $4}') us/turn"
  done
done
