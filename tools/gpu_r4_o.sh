#!/bin/bash
# Round 4, visit o: the sizing pass as ONE resident round (6144 workgroups), per-kind lists, dealt shares -- pre-filter on (the
# pipeline's conditions) and off (kbench), against the build before; parity
OUT=gpurun_out/r04o
mkdir -p $OUT
{
echo "-- pre-filter on (tools/slice_pf_bench.py): before / now / now even shares / now one launch / grids 5120, 7168"
python tools/slice_pf_bench.py rtl_433_amd/lib/ab/v0_dense.so 8 0 2>&1 | tail -1
python tools/slice_pf_bench.py rtl_433_amd/lib/librtl433hip.so 8 0 2>&1 | tail -1
python tools/slice_pf_bench.py rtl_433_amd/lib/librtl433hip.so 8 524288 2>&1 | tail -1
python tools/slice_pf_bench.py rtl_433_amd/lib/librtl433hip.so 8 131072 2>&1 | tail -1
for g in 5120 7168; do R433_SLICE_GRID=$g python tools/slice_pf_bench.py rtl_433_amd/lib/librtl433hip.so 8 0 2>&1 | tail -1; done
echo "-- no pre-filter (tools/variant_bench.py): before / now, 8192 and 1024 captures"
python tools/variant_bench.py rtl_433_amd/lib/ab/v0_dense.so 8192 8 1 0 2>&1 | tail -1
python tools/variant_bench.py rtl_433_amd/lib/librtl433hip.so 8192 8 1 0 2>&1 | tail -1
python tools/variant_bench.py rtl_433_amd/lib/ab/v0_dense.so 1024 8 1 0 2>&1 | tail -1
python tools/variant_bench.py rtl_433_amd/lib/librtl433hip.so 1024 8 1 0 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
echo "== who the pass waits for"
R433_SLICE_TICKS=1 timeout 300 python tools/kbench.py --reps 4 --streams 8192 2>&1 | grep "r.slice: small" | tail -13 | tee $OUT/ticks.txt
echo "== parity"
timeout 900 python -m pytest tests -m gpu -q -x -k "ragged or slicer or full_size or fuzz or prefilter or shard or dispatch" 2>&1 | tail -2
timeout 600 python tools/fuzz_emu.py --gpu 900 530000 2>&1 | tail -1
