#!/bin/bash
# Round 4, visit l: the sizing pass -- the build before (a slot per device, even shares, every chunk walks every package)
# against per-kind lists + dealt shares, same box
OUT=gpurun_out/r04l
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "-- before (rtl_433_amd/lib/ab/v0_dense.so)"
timeout 300 python tools/variant_bench.py rtl_433_amd/lib/ab/v0_dense.so 8192 8 1 0 2>&1 | tail -1
echo "-- now, even shares (R433_DEBUG_EVEN_SLICE)"
timeout 300 python tools/variant_bench.py rtl_433_amd/lib/librtl433hip.so 8192 8 1 524288 2>&1 | tail -1
echo "-- now, shares by measured work"
timeout 300 python tools/variant_bench.py rtl_433_amd/lib/librtl433hip.so 8192 8 1 0 2>&1 | tail -1
echo "-- now, one launch (R433_DEBUG_ONE_SLICE_LAUNCH), shares by measured work"
timeout 300 python tools/variant_bench.py rtl_433_amd/lib/librtl433hip.so 8192 8 1 131072 2>&1 | tail -1
echo "-- 1024 captures: before / now"
timeout 300 python tools/variant_bench.py rtl_433_amd/lib/ab/v0_dense.so 1024 8 1 0 2>&1 | tail -1
timeout 300 python tools/variant_bench.py rtl_433_amd/lib/librtl433hip.so 1024 8 1 0 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
echo "== who the pass waits for (shares by measured work, fourth run)"
R433_SLICE_TICKS=1 timeout 300 python tools/kbench.py --reps 4 --streams 8192 2>&1 | grep "r.slice: small" | tail -13 | tee $OUT/ticks.txt
