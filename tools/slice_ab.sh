# A/B of the slicers' sizing pass on the GPU box: packages drawn heaviest first from a cursor per chunk of devices (default)
# against fixed strides (R433_DEBUG_STATIC_SLICE); prints kernel times and digests of the package / event records
L=rtl_433_amd/lib/librtl433hip.so
for n in ${@:-8192}; do for dbg in 0 65536; do timeout 40 python tools/variant_bench.py $L $n 4 1 $dbg 2>&1 | grep -v amdgpu.ids | tail -1; done; done | tee gpurun_out/slice_ab.txt
