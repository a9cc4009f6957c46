{
for n in 8192 4096 2048 1536 1024; do echo "== kbench $n captures: single / pair"; for f in 4096 32768; do timeout 300 python tools/kbench.py --nodevs --reps 4 --streams $n --debug $f 2>&1 | tail -1 | cut -c1-70; done; done
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_seam_lib.py tests/test_seam_functions.py -m gpu -x -q 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids
