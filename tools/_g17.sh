mkdir -p gpurun_out/r02_n
{
for f in 0 8192 16384 24576 4096; do echo "== kbench debug $f"; timeout 300 python tools/kbench.py --nodevs --reps 7 --debug $f 2>&1 | tail -2; done
echo "== fsk-cu8"; for f in 0 8192 16384 24576; do timeout 300 python tools/kbench.py --nodevs --fsk-cu8 --debug $f 2>&1 | tail -1; done
echo "== cs16"; for f in 0 24576; do timeout 300 python tools/kbench.py --nodevs --cs16 --debug $f 2>&1 | tail -1; done
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_n/out.txt
