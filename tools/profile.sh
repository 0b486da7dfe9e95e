#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + HBM traffic + SQ counters for one bench shape.
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass, and
# --pmc is never combined with other trace domains).
#   tools/profile.sh <round tag> <key> [bench args...]
# <key> names the shape (c2, ww8_1m, c4, c3, c2_k1 ...): summaries go to gpurun_out/profiles_out/
# (<tag>_<key>_kernel_stats.csv, <tag>_<key>_counters.json, pmc_<key>.json); copy them into profiles/.
set -u
TAG=${1:-r02}; KEY=${2:-c2}; shift 2 || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_${KEY}
mkdir -p "$OUT"
B="--no-cpu-baseline --no-other-shapes --no-from-init --no-unfused"
# <key>_k1: the single-turn (max_fuse = 1) kernel of the shape in its sustained regime - the plain bench line's hbm_streaming
# part replays hipGraphs of back-to-back single-turn launches; pmc_summary.py picks the ge_step_kernel<.., true> rows
case "$KEY" in *_k1) B="--no-cpu-baseline --no-other-shapes --no-from-init --steps 2 --warmup 0" ;; esac
# warm the GPU up first: on a fresh box the first launch is taken at idle clocks (~1.56 ms instead of ~1.13 for C2)
# and skews the kernel-trace average
python3 bench.py $B --steps 2 --warmup 1 "$@" > /dev/null 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py $B "$@" > "$OUT/bench_kt.json" 2> "$OUT/kt.err" || echo "kt failed"
echo "[$KEY] kernel-trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $B "$@" > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err" || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py $B "$@" > "$OUT/bench_write.json" 2> "$OUT/write.err" || echo "write failed"
echo "[$KEY] traffic passes done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -- python3 bench.py $B "$@" > "$OUT/bench_sq.json" 2> "$OUT/sq.err" || echo "sq failed"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d "$OUT/sq2" -- python3 bench.py $B "$@" > "$OUT/bench_sq2.json" 2> "$OUT/sq2.err" || echo "sq2 failed"
echo "[$KEY] SQ passes done"
python3 tools/pmc_summary.py "$OUT" "$TAG" "$KEY"
