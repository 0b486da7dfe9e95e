#!/bin/bash
# After `tools/gpu_baseline.sh <tag>` on a GPU box: copy what it left under gpurun_out/ into profiles/ under the round's
# names (r03_*, pmc_*.json, the three bench records), and check that every counter profile carries the hash of the device
# code in this tree; then rewrite the measurement tables of the docs from them.   tools/adopt_profiles.sh <tag> [round-prefix, default r03]
set -eu
TAG=$1; ROUND=${2:-r05}
cd "$(dirname "$0")/.."
for f in gpurun_out/profiles_out/${TAG}_*; do cp "$f" "profiles/${ROUND}_$(basename "${f#gpurun_out/profiles_out/${TAG}_}")"; done
cp gpurun_out/profiles_out/pmc_*.json profiles/
cp gpurun_out/$TAG/bench_n1.json profiles/${ROUND}_bench_n1_driver_flags.json
cp gpurun_out/$TAG/bench_n2_gloo.json profiles/${ROUND}_bench_n2_gloo_rehearsal.json
cp gpurun_out/$TAG/bench_rccl_one_rank.json profiles/${ROUND}_bench_rccl_one_rank.json
[ -s gpurun_out/$TAG/bench_c5.json ] && cp gpurun_out/$TAG/bench_c5.json profiles/${ROUND}_bench_c5_share.json
python - <<'PY'
import glob, json, sys
sys.path.insert(0, ".")
from game_engine_amd._lib import kernel_source_hash
h = kernel_source_hash()
stale = [p for p in sorted(glob.glob("profiles/pmc_*.json")) if json.load(open(p)).get("kernel_src_sha256") not in (None, h) or "kernel_src_sha256" not in json.load(open(p))]
stale = [p for p in stale if "traffic" not in p]
print("device-code hash", h[:12], "- stale profiles:", stale or "none")
PY
make -C game_engine_amd/csrc asm -s > /dev/null 2>&1; python tools/valu_mix.py --write      # the mean instruction price per fused kernel, tied to the device-code hash
python tools/design_tables.py $ROUND --write      # the marked tables / values of DESIGN.md, BASELINE.md, profiles/README.md
