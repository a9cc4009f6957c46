mkdir -p gpurun_out/r02_p
{
for n in 1 256 512 1024; do for f in 0 4096; do echo "== streams $n (worst capture first) debug $f"; timeout 300 python tools/kbench.py --nodevs --reps 7 --streams $n --seed0 885 --debug $f 2>&1 | tail -1 | cut -c1-90; done; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_p/out.txt
