export TMPDIR=/tmp
OUT=gpurun_out/r05_k; mkdir -p $OUT
timeout 170 python tools/split_quiet_stats.py 16 </dev/null 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-400 | tee $OUT/split_quiet.txt
