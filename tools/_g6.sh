mkdir -p gpurun_out/r02_f
echo "== bench default (steps 30)"; timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/r02_f/bench_c2.json 2> gpurun_out/r02_f/bench_c2.err; tail -3 gpurun_out/r02_f/bench_c2.err; cat gpurun_out/r02_f/bench_c2.json
echo "== bench config 3"; timeout 900 python bench.py --config 3 --steps 3 --warmup 1 > gpurun_out/r02_f/bench_c3.json 2> gpurun_out/r02_f/bench_c3.err; tail -3 gpurun_out/r02_f/bench_c3.err; cat gpurun_out/r02_f/bench_c3.json
