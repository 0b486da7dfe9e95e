#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + HBM traffic counters for the bench command.
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass, and
# --pmc is never combined with other trace domains).  Usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --no-cpu-baseline --no-other-shapes --no-from-init --no-unfused "$@" > "$OUT/bench_kt.json" 2> "$OUT/kt.err" || echo "kt failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --no-cpu-baseline --no-other-shapes --no-from-init --no-unfused "$@" > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err" || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --no-cpu-baseline --no-other-shapes --no-from-init --no-unfused "$@" > "$OUT/bench_write.json" 2> "$OUT/write.err" || echo "write failed"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -- python3 bench.py --no-cpu-baseline --no-other-shapes --no-from-init --no-unfused "$@" > "$OUT/bench_sq.json" 2> "$OUT/sq.err" || echo "sq failed"
python3 tools/pmc_summary.py "$OUT" "$TAG"
