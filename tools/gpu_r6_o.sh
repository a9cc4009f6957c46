#!/bin/bash
# round 6, visit o: the small kernels of a pass after k_pkg_order went to 1024 threads and the slice index to eight loads in flight:
# one GPU leg with the real decoders' pre-filter tables (tools/slice_pf_bench.py), then the tests that walk the index
TAG=${1:-r06_o}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
{ for i in 1 2 3; do timeout 300 python tools/slice_pf_bench.py </dev/null 2>&1 | tail -1; done; } | grep -v amdgpu.ids | cut -c1-330 | tee $OUT/leg.txt
timeout 600 python -m pytest tests/test_dispatch.py tests/test_prefilter.py -m gpu -x -q </dev/null 2>&1 | tail -3 | tee $OUT/pytest_dispatch.txt
