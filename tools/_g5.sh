export TMPDIR=/tmp
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  tag=r02_e_$(echo $pmc | cut -d' ' -f1)
  bash tools/pmc_run.sh $tag "$pmc" --nodevs --streams 1 --seed0 885 2>&1 | tail -5
done
