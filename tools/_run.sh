mkdir -p gpurun_out/r03o
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03o/bench_n1.json 2> gpurun_out/r03o/bench_n1.err; echo "bench rc=$?"
GE_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/r03o/bench_n2_gloo.json 2> gpurun_out/r03o/bench_n2.err; echo "bench2 rc=$?"
