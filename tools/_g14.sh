mkdir -p gpurun_out/r02_l
{
echo "== pytest new gpu tests"; timeout 1200 python -m pytest tests/test_logic_dump.py tests/test_dropin.py tests/test_dispatch.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3
echo "== kbench"; python tools/kbench.py --nodevs --reps 7 2>&1 | tail -2
echo "== kbench timing"; python tools/kbench.py --nodevs --debug 1024 2>&1 | tail -14 | head -10
echo "== kbench cs16"; python tools/kbench.py --nodevs --cs16 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_l/out.txt
