#!/bin/bash
# round 4, first visit: kernel A/B (round-3 build vs lazy tiles), GPU parity suite, GPU fuzz
OUT=gpurun_out/r4a
mkdir -p $OUT
export TMPDIR=/tmp
echo "== variant bench (8192 captures, no decoders)" | tee $OUT/variants.txt
for lib in rtl_433_amd/lib/librtl433hip_r3.so rtl_433_amd/lib/librtl433hip.so; do
  echo "-- $lib" | tee -a $OUT/variants.txt
  timeout 300 python tools/variant_bench.py $lib 8192 6 0 2>&1 | tail -1 | tee -a $OUT/variants.txt
done
echo "-- lazy off by flag" | tee -a $OUT/variants.txt
timeout 300 python tools/variant_bench.py rtl_433_amd/lib/librtl433hip.so 8192 6 0 262144 2>&1 | tail -1 | tee -a $OUT/variants.txt
echo "-- 1024 captures" | tee -a $OUT/variants.txt
timeout 300 python tools/variant_bench.py rtl_433_amd/lib/librtl433hip_r3.so 1024 6 0 2>&1 | tail -1 | tee -a $OUT/variants.txt
timeout 300 python tools/variant_bench.py rtl_433_amd/lib/librtl433hip.so 1024 6 0 2>&1 | tail -1 | tee -a $OUT/variants.txt
echo "== gpu fuzz 600" | tee $OUT/fuzz.txt
timeout 900 python tools/fuzz_emu.py --gpu 600 100000 2>&1 | tail -8 | tee -a $OUT/fuzz.txt
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest.txt
