export TMPDIR=/tmp
t0=$(date +%s)
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  now=$(date +%s); [ $((now - t0)) -gt 55 ] && break
  timeout 40 bash tools/pmc_run.sh r03_slice_$(echo $pmc | cut -d' ' -f2) "$pmc" 2>&1 | grep -v amdgpu.ids | cut -c1-140
done | tee gpurun_out/r03_slice_pmc.txt
