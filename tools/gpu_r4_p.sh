#!/bin/bash
# Round 4, visit p: the sizing pass after the fork-order fix, floors, every workgroup measured
OUT=gpurun_out/r04p
mkdir -p $OUT
echo "== parity first"
timeout 600 python tools/fuzz_emu.py --gpu 700 550000 2>&1 | tail -1 | tee $OUT/fuzz.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "ragged or slicer or prefilter" 2>&1 | tail -2 | tee $OUT/pytest.txt
{
echo "-- pre-filter on (tools/slice_pf_bench.py, 12 runs): before / now / now one launch"
python tools/slice_pf_bench.py rtl_433_amd/lib/ab/v0_dense.so 12 0 2>&1 | tail -1
python tools/slice_pf_bench.py rtl_433_amd/lib/librtl433hip.so 12 0 2>&1 | tail -1
python tools/slice_pf_bench.py rtl_433_amd/lib/librtl433hip.so 12 131072 2>&1 | tail -1
echo "-- no pre-filter (tools/variant_bench.py): before / now, 8192, 4096, 1024 captures"
for n in 8192 1024; do
python tools/variant_bench.py rtl_433_amd/lib/ab/v0_dense.so $n 10 1 0 2>&1 | tail -1
python tools/variant_bench.py rtl_433_amd/lib/librtl433hip.so $n 10 1 0 2>&1 | tail -1
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
