export TMPDIR=/tmp
TAG=${1:-r05_i}; OUT=gpurun_out/$TAG; mkdir -p $OUT
echo "== the bench's own batch (every third capture a protocol transmission): default / capture order / pairs"
{ timeout 150 python tools/slice_pf_bench.py "" 8 0 </dev/null 2>&1 | tail -1
  timeout 150 python tools/slice_pf_bench.py "" 8 64 </dev/null 2>&1 | tail -1
  timeout 150 python tools/slice_pf_bench.py "" 8 8388608 </dev/null 2>&1 | tail -1; } | grep -v amdgpu.ids | tee $OUT/spb.txt
