mkdir -p gpurun_out/r02_o
{
for f in 0 4096; do echo "== kbench debug $f"; timeout 300 python tools/kbench.py --nodevs --reps 7 --debug $f 2>&1 | tail -2; done
echo "== fsk-cu8"; timeout 300 python tools/kbench.py --nodevs --fsk-cu8 2>&1 | tail -1
echo "== cs16"; timeout 300 python tools/kbench.py --nodevs --cs16 2>&1 | tail -1
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_slicer_matrix.py -m gpu -x -q 2>&1 | tail -2
echo "== fuzz gpu 1000"; timeout 600 python tools/fuzz_emu.py --gpu 1000 52000 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_o/out.txt
