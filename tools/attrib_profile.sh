#!/bin/bash
# Where does the vector pipe's idle time go?  Counter passes beside tools/profile.sh's (never combined with a trace domain):
# LDS pipe (array cycles, bank / address conflicts, loads / stores / atomics), front end (instruction fetches, I-cache misses,
# branches, resident waves), lane use (SQ_THREAD_CYCLES_VALU) and the wait buckets.
#   tools/attrib_profile.sh <round tag> <key> [bench args...]     -> gpurun_out/profiles_out/<tag>_<key>_counters.json
set -u
TAG=${1:-r05}; KEY=${2:-ww8_1048576}; shift 2 || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_${KEY}
mkdir -p "$OUT"
B="--no-cpu-baseline --no-other-shapes --no-from-init --no-unfused"
case "$KEY" in *_k1) B="--no-cpu-baseline --no-other-shapes --no-from-init --steps 2 --warmup 0" ;; esac
python3 bench.py $B --steps 2 --warmup 1 "$@" > /dev/null 2>&1 || true
run() { sub=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$sub" -- python3 bench.py $B $ARGS > "$OUT/bench_$sub.json" 2> "$OUT/$sub.err" || echo "$sub failed"; }
ARGS="$*"
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU
run sq3 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
run sq4 SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_BUSY_CU_CYCLES
run sq5 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES
echo "[$KEY] attribution passes done"
python3 tools/pmc_summary.py "$OUT" "$TAG" "${KEY}_attrib" > /dev/null
python3 - <<PY
import json
d = json.load(open("gpurun_out/profiles_out/${TAG}_${KEY}_attrib_counters.json"))
print({k: round(v["per_wave_turn"], 2) for k, v in d.items() if isinstance(v, dict) and "per_wave_turn" in v})
PY
