mkdir -p gpurun_out/r02_m
{
echo "== parity (gpu subset)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_slicer_matrix.py tests/test_logic_dump.py -m gpu -x -q 2>&1 | tail -4
echo "== kbench P/C"; timeout 300 python tools/kbench.py --nodevs --reps 7 2>&1 | tail -2
echo "== kbench one-wave (debug 4096)"; timeout 300 python tools/kbench.py --nodevs --reps 7 --debug 4096 2>&1 | tail -2
echo "== kbench rotate 3 P/C"; timeout 300 python tools/kbench.py --nodevs --reps 7 --rotate 3 2>&1 | tail -2
echo "== kbench cs16"; timeout 300 python tools/kbench.py --nodevs --cs16 2>&1 | tail -1; timeout 300 python tools/kbench.py --nodevs --cs16 --debug 4096 2>&1 | tail -1
echo "== kbench fsk-cu8"; timeout 300 python tools/kbench.py --nodevs --fsk-cu8 2>&1 | tail -1; timeout 300 python tools/kbench.py --nodevs --fsk-cu8 --debug 4096 2>&1 | tail -1
echo "== fuzz gpu 2000"; timeout 600 python tools/fuzz_emu.py --gpu 2000 20000 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_m/out.txt
