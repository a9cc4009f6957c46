for i in 1 2; do timeout 600 python bench.py --quick --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['breakdown_ms'])"; done
bash tools/gpu_round.sh r02_j 2>&1 | grep -A12 "rocprof kernel trace" | head -8
cat gpurun_out/r02_j/prof_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('profiled run:', d['value'], d['roofline'], d['breakdown_ms'])"
