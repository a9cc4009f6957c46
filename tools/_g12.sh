mkdir -p gpurun_out/r02_k
{
echo "== pytest dispatch/dropin"; timeout 900 python -m pytest tests/test_dispatch.py tests/test_dropin.py -m gpu -q 2>&1 | tail -3
echo "== cli bench"; timeout 600 tools/cli_bench.sh 1024 gpurun_out/r02_k 2>&1 | tail -4
echo "== kbench same input"; python tools/kbench.py --nodevs --reps 9 2>&1 | tail -2
echo "== kbench rotate 3"; python tools/kbench.py --nodevs --reps 9 --rotate 3 2>&1 | tail -2
echo "== kbench A+B only (debug 256)"; python tools/kbench.py --nodevs --reps 5 --debug 256 2>&1 | tail -1
echo "== bench"; timeout 600 python bench.py --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['breakdown_ms']); print(json.dumps(d['real_decoders']))"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_k/out.txt
