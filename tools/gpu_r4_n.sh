#!/bin/bash
# Round 4, visit n: the slicers with the pre-filter on (what the pipeline runs): the build before, the new one, grid sizes
OUT=gpurun_out/r04n
mkdir -p $OUT
{
python tools/slice_pf_bench.py rtl_433_amd/lib/ab/v0_dense.so 8 0 2>&1 | tail -1
python tools/slice_pf_bench.py rtl_433_amd/lib/librtl433hip.so 8 524288 2>&1 | tail -1
python tools/slice_pf_bench.py rtl_433_amd/lib/librtl433hip.so 8 0 2>&1 | tail -1
python tools/slice_pf_bench.py rtl_433_amd/lib/librtl433hip.so 8 131072 2>&1 | tail -1
for g in 12288 8192 6144 4096; do R433_SLICE_GRID=$g python tools/slice_pf_bench.py rtl_433_amd/lib/librtl433hip.so 8 0 2>&1 | tail -1; done
R433_SLICE_GRID=8192 python tools/slice_pf_bench.py rtl_433_amd/lib/librtl433hip.so 8 524288 2>&1 | tail -1
R433_SLICE_GRID=8192 python tools/slice_pf_bench.py rtl_433_amd/lib/ab/v0_dense.so 8 0 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
