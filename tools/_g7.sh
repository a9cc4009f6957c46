mkdir -p gpurun_out/r02_g
for c in 3 5 4; do
echo "== bench config $c"; timeout 1200 python bench.py --config $c > gpurun_out/r02_g/bench_c$c.json 2> gpurun_out/r02_g/bench_c$c.err; tail -3 gpurun_out/r02_g/bench_c$c.err; cat gpurun_out/r02_g/bench_c$c.json
done
