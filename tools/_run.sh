mkdir -p gpurun_out/r03e
GE_LIB_PATH=$PWD/game_engine_amd/ab/kp.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "batch_equals_oracle or restart" 2>&1 | tail -1
tools/abn.sh "ww:8:65536 ww:8:1048576 ww:12:2097152 tt:4:1048576" game_engine_amd/ab/reorder.so game_engine_amd/ab/kp.so > gpurun_out/r03e/ab.txt 2>&1
grep -c fuse gpurun_out/r03e/ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-other-shapes --no-from-init --no-unfused"
for key in "c4 --workload c4" "ww8_1048576 --rooms 1048576" "c3 --workload c3" "c2"; do set -- $key; k=$1; shift
  mkdir -p gpurun_out/prof_r03e_$k
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d gpurun_out/prof_r03e_$k/sq -- python3 bench.py $B --steps 2 --warmup 0 "$@" > gpurun_out/prof_r03e_$k/bench_sq.json 2> gpurun_out/prof_r03e_$k/sq.err || echo "sq failed"
  python3 tools/pmc_summary.py gpurun_out/prof_r03e_$k r03e $k | grep -A4 instructions_per
done
