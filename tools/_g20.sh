mkdir -p gpurun_out/r02_q
{
for f in 1024 5120; do echo "== worst capture alone, debug $f"; timeout 300 python tools/kbench.py --nodevs --reps 5 --streams 1 --seed0 885 --debug $f 2>&1 | tail -19 | cut -c1-110; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_q/out.txt
