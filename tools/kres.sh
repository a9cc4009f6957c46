#!/bin/bash
# Register / scratch / LDS use of every kernel of one source file, one line per kernel:
#   tools/kres.sh rtl_433_amd/csrc/stream_kernels.hip [extra hipcc flags]
src=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$(dirname "$0")/../include" -I"$(dirname "$0")/../rtl_433_amd/csrc" "$@" -c "$src" \
    -Rpass-analysis=kernel-resource-usage -o /dev/null 2>&1 |
awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {l=$(NF-1); print name, "vgpr", v, "scratch", s, "occ", o, "lds", l}' |
sed -E 's/ \[-Rpass[^]]*\]//g' | while read n rest; do echo "$(echo "$n" | c++filt | sed -E 's/r433::\(anonymous namespace\):://; s/\(.*//') $rest"; done
