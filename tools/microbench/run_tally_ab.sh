#!/bin/bash
# Builds (in the container: hipcc cross-compiles) and runs (on the GPU box) the vote-tally mapping A/B.
set -e
cd "$(dirname "$0")"
[ -x tally_ab ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tally_ab tally_ab.hip
./tally_ab
