#!/bin/bash
# One GPU-box call: the -m gpu suite, the bench under the driver's flags, C2 counters.  tools/gpu_check.sh <tag>
set -u
TAG=${1:-r02}
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/$TAG/pytest.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_n1.json 2> gpurun_out/$TAG/bench_n1.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/$TAG/bench_n1.json").read().strip().splitlines()[-1])
print("value %.4g frac %.4f ms/step %.4f launch_us %.1f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"], d["roofline"]["avg_launch_us"]))
print("issue", d["issue"].get("frac"), "streaming", d["hbm_streaming"] and d["hbm_streaming"]["frac_wall"])
for k,v in (d.get("other_shapes") or {}).items(): print(k, "%.4g" % v["value"], "us/turn %.3f" % v["us_per_turn"])
PY
bash tools/profile.sh $TAG c2 --steps 4 --warmup 1 | tail -12
