#!/bin/bash
# kernel-trace of single-turn launch chains (GE_CHAINS = 1 / 2 / 4): do the chains' launches overlap on the device?
#   tools/chains_trace.sh <tag> "ww:8:1048576" 67108864
set -u
TAG=${1:-r05}; SHAPE=${2:-ww:8:1048576}; BYTES=${3:-67108864}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/k1_probe.py $SHAPE > /dev/null 2>&1 || true      # warm the box
for ch in 1 2 4; do
  OUT=gpurun_out/chains_${TAG}_$ch
  GE_CHAINS=$ch rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 tools/k1_probe.py $SHAPE 2> "$OUT.err" | tail -1
  echo -n "GE_CHAINS=$ch  "; python3 tools/trace_overlap.py "$OUT" $BYTES
done
