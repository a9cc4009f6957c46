mkdir -p gpurun_out/r02_d
{
echo "== parity (gpu subset)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_slicer_matrix.py -m gpu -x -q 2>&1 | tail -4
echo "== fuzz gpu 3000"; timeout 600 python tools/fuzz_emu.py --gpu 3000 20000 2>&1 | tail -3
echo "== kbench engine"; python tools/kbench.py --nodevs 2>&1 | tail -1
echo "== kbench engine timing"; python tools/kbench.py --nodevs --debug 1024 2>&1 | tail -14 | head -10
echo "== kbench cs16"; python tools/kbench.py --nodevs --cs16 2>&1 | tail -1
echo "== kbench fsk-cu8"; python tools/kbench.py --nodevs --fsk-cu8 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_d/kbench.txt
