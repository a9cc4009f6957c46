export TMPDIR=/tmp
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_INSTS_SENDMSG"; do
  tag=r02_b_$(echo $pmc | cut -d' ' -f1)
  bash tools/pmc_run.sh $tag "$pmc" --nodevs --streams 1 --seed0 885 2>&1 | tail -6
done
python tools/kbench.py --nodevs --streams 1 --seed0 885 --debug 1024 | tail -12
