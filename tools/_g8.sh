mkdir -p gpurun_out/r02_h
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r02_h/pytest.txt
echo "== bench threads"; for t in 32 64 128; do timeout 600 python bench.py --quick --steps 60 --threads $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($t, d['value'], d['ms_per_step'], d['breakdown_ms'])"; done | tee gpurun_out/r02_h/threads.txt
