export TMPDIR=/tmp
for dbg in 0 256; do
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=r02_r_${dbg}_$(echo $pmc | cut -d' ' -f2)
  echo "== debug $dbg: $pmc"
  bash tools/pmc_run.sh $tag "$pmc" --nodevs --streams 1 --seed0 885 --debug $dbg 2>&1 | grep "ELb1ELb0ELb0" | cut -c40-120
done; done
