#!/bin/bash
# The round's GPU records, in four box calls (a call is limited to 20 minutes):
#   tools/gpu_baseline.sh <tag> fused     counter / kernel-trace passes of the fused bench shapes
#   tools/gpu_baseline.sh <tag> attrib    the attribution passes of the same shapes (tools/attrib_profile.sh: LDS, front end, lanes, CU-busy cycles)
#   tools/gpu_baseline.sh <tag> single    ... of the single-turn launches, BASELINE shapes and the beyond-Infinity-Cache shapes
#   (copy gpurun_out/profiles_out/pmc_*.json into profiles/ in between: the bench lines quote the profiles of THIS kernel build)
#   tools/gpu_baseline.sh <tag> bench     the -m gpu suite, the bench under the driver's flags, the C5 line, the 2-rank rehearsal, RCCL with one rank
# then tools/adopt_profiles.sh <tag> <round>.
set -u
TAG=${1:-r05}; PART=${2:-bench}
mkdir -p gpurun_out/$TAG
case "$PART" in
fused)
  bash tools/profile.sh $TAG c2 --steps 4 --warmup 1
  bash tools/profile.sh $TAG ww8_1048576 --rooms 1048576 --steps 2 --warmup 0
  bash tools/profile.sh $TAG c4 --workload c4 --steps 2 --warmup 0
  bash tools/profile.sh $TAG c3 --workload c3 --steps 2 --warmup 0
  bash tools/profile.sh $TAG c5 --workload c5 --steps 2 --warmup 0        # one GPU's share of the mixed batch (ge_step_kernel_mixed)
  ;;
attrib)
  bash tools/attrib_profile.sh $TAG c2 --steps 4 --warmup 1
  bash tools/attrib_profile.sh $TAG ww8_1048576 --rooms 1048576 --steps 2 --warmup 0
  bash tools/attrib_profile.sh $TAG c4 --workload c4 --steps 2 --warmup 0
  bash tools/attrib_profile.sh $TAG c3 --workload c3 --steps 2 --warmup 0
  bash tools/attrib_profile.sh $TAG c5 --workload c5 --steps 2 --warmup 0
  ;;
single)
  bash tools/profile.sh $TAG c2_k1
  bash tools/profile.sh $TAG ww8_1048576_k1 --rooms 1048576
  bash tools/profile.sh $TAG c4_k1 --workload c4
  bash tools/profile.sh $TAG c3_k1 --workload c3
  bash tools/profile.sh $TAG c5_k1 --workload c5                          # the mixed kernel's single-turn build
  # resident state larger than the 256 MiB Infinity Cache: the HBM figure that is provably HBM
  bash tools/profile.sh $TAG ww8_33554432_k1 --workload ww8_33554432
  bash tools/profile.sh $TAG c4_whole_k1 --workload c4_whole
  bash tools/profile.sh $TAG tt4_33554432_k1 --workload tt4_33554432
  ;;
bench)
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest.log
  timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_n1.json 2> gpurun_out/$TAG/bench_n1.err; echo "bench rc=$?"
  python - <<PY
import json
d=json.loads(open("gpurun_out/$TAG/bench_n1.json").read().strip().splitlines()[-1])
print("value %.4g frac %.4f ms/step %.4f launch_us %.1f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["roofline"]["avg_launch_us"]))
print("issue", d["issue"].get("frac"), "streaming", d["hbm_streaming"] and d["hbm_streaming"]["frac"])
for k, v in (d.get("other_shapes") or {}).items():
    print("%-46s %.3g steps/s, single-turn %.3f of 8 TB/s" % (k, v["value"], v["hbm_streaming"]["frac"]))
for k, v in (d.get("hbm_streaming_beyond_l3") or {}).items():
    print("%-62s single-turn %.3f of 8 TB/s, parity %s" % (k, v["frac"], v["parity"]["single_turn_equals_fused"]))
PY
  timeout -k 10 300 python bench.py --gpus 1 --workload c5 --steps 8 --warmup 2 --no-cpu-baseline --no-other-shapes --no-from-init > gpurun_out/$TAG/bench_c5.json 2> gpurun_out/$TAG/bench_c5.err; echo "bench c5 rc=$?"
  GE_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/$TAG/bench_n2_gloo.json 2> gpurun_out/$TAG/bench_n2.err; echo "bench2 rc=$?"
  GE_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-other-shapes --no-from-init --no-unfused > gpurun_out/$TAG/bench_rccl_one_rank.json 2> gpurun_out/$TAG/bench_rccl.err; echo "bench rccl rc=$?"
  ;;
esac
# the raw per-pass directories have been summarised into gpurun_out/profiles_out/ (tools/pmc_summary.py); they alone exceed what a box call
# copies back (64 MiB)
case "$PART" in fused|attrib|single) rm -rf gpurun_out/prof_${TAG}_* ;; esac
