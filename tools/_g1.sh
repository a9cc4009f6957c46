mkdir -p gpurun_out/r02_a
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r02_a/pytest.txt
echo "== cli bench"; timeout 600 tools/cli_bench.sh 1024 gpurun_out/r02_a 2>&1 | tail -5
echo "== kbench"; for a in "" "--sigma-only" "--nodevs --debug 1024" "--nodevs --sigma-only --debug 1024"; do timeout 300 python tools/kbench.py $a 2>&1 | tail -16; done | tee gpurun_out/r02_a/kbench.txt
