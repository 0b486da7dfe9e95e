#!/bin/bash
# One GPU-box call: the -m gpu suite, the bench under the driver's flags, the 2-rank rehearsal, and
# counter passes for the bench shapes (fused and single-turn).  tools/gpu_baseline.sh <tag>
set -u
TAG=${1:-r03}
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest.log
# counter passes first: the bench lines below quote the profiles of THIS kernel build (profiles/pmc_*.json carry the kernel
# sources' hash; the copy below only lives on the box - copy gpurun_out/profiles_out/ into profiles/ by hand afterwards)
bash tools/profile.sh $TAG c2 --steps 4 --warmup 1
bash tools/profile.sh $TAG ww8_1048576 --rooms 1048576 --steps 2 --warmup 0
bash tools/profile.sh $TAG c4 --workload c4 --steps 2 --warmup 0
bash tools/profile.sh $TAG c3 --workload c3 --steps 2 --warmup 0
bash tools/profile.sh $TAG c5 --workload c5 --steps 2 --warmup 0        # one GPU's share of the mixed batch (ge_step_kernel_mixed)
bash tools/profile.sh $TAG c2_k1
bash tools/profile.sh $TAG ww8_1048576_k1 --rooms 1048576
bash tools/profile.sh $TAG c4_k1 --workload c4
bash tools/profile.sh $TAG c3_k1 --workload c3
cp gpurun_out/profiles_out/pmc_*.json profiles/
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_n1.json 2> gpurun_out/$TAG/bench_n1.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/$TAG/bench_n1.json").read().strip().splitlines()[-1])
print("value %.4g frac %.4f ms/step %.4f launch_us %.1f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["roofline"]["avg_launch_us"]))
print("issue", d["issue"].get("frac"), "streaming", d["hbm_streaming"] and d["hbm_streaming"]["frac"])
for k, v in (d.get("other_shapes") or {}).items():
    print("%-46s %.3g steps/s, single-turn %.3f of 8 TB/s" % (k, v["value"], v["hbm_streaming"]["frac"]))
PY
timeout -k 10 300 python bench.py --gpus 1 --workload c5 --steps 8 --warmup 2 --no-cpu-baseline --no-other-shapes --no-from-init > gpurun_out/$TAG/bench_c5.json 2> gpurun_out/$TAG/bench_c5.err; echo "bench c5 rc=$?"
GE_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/$TAG/bench_n2_gloo.json 2> gpurun_out/$TAG/bench_n2.err; echo "bench2 rc=$?"
GE_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-other-shapes --no-from-init --no-unfused > gpurun_out/$TAG/bench_rccl_one_rank.json 2> gpurun_out/$TAG/bench_rccl.err; echo "bench rccl rc=$?"
