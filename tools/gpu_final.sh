#!/bin/bash
# The closing GPU visit of a round: everything profiles/ quotes, from one build.  tools/gpu_final.sh <tag>
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_round.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -30
echo "== kbench (kernel alone; producer / consumer pair vs one wavefront, by grid size)"
{ for n in 1024 3072 8192; do for f in 32768 4096; do timeout 300 python tools/kbench.py --nodevs --reps 7 --streams $n --debug $f 2>&1 | tail -1; done; done
  timeout 300 python tools/kbench.py --reps 5 2>&1 | tail -1
  timeout 300 python tools/kbench.py --reps 4 --streams 8192 2>&1 | tail -1
  timeout 300 python tools/kbench.py --nodevs --cs16 2>&1 | tail -1
  timeout 300 python tools/kbench.py --nodevs --fsk-cu8 2>&1 | tail -1; } 2>&1 | grep -v amdgpu.ids | tee $OUT/kbench.txt
echo "== PMC traffic"
timeout 900 python tools/pmc_traffic.py 2>&1 | tail -4
echo "== PMC SQ (instruction mix of one launch of the bench batch)"
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  bash tools/pmc_run.sh ${TAG}_sq_$(echo $pmc | cut -d' ' -f2) "$pmc" --nodevs --streams 8192 2>&1 | grep k_wave | cut -c1-130
done | tee $OUT/pmc_sq.txt
echo "== configs 3 / 4 / 5"
for c in 3 4 5; do timeout 900 python bench.py --config $c > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; python -c "
import json,sys
d=json.load(open('$OUT/bench_c$c.json')); print('config $c:', d['value'], d['unit'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('parity'))"; done
echo "== CLI drop-in"
timeout 600 tools/cli_bench.sh 1024 $OUT 2>&1 | tail -3
echo "== the C pipeline host"
timeout 300 bash tools/pipeline_host_bench.sh 8192 2>&1 | tail -9 | tee $OUT/pipeline_host.txt
echo "== probes: PCIe link, a pass beside a copy, capture order inside a grid"
{ timeout 120 python tools/pcie_probe.py; timeout 120 python tools/overlap_probe.py; timeout 120 python tools/order_probe.py 8192; } 2>&1 | grep -v amdgpu.ids | tee $OUT/probes.txt
echo "== fuzz (GPU, 8000 cases)"
timeout 1500 python tools/fuzz_emu.py --gpu 8000 90000 2>&1 | tail -1 | tee $OUT/fuzz.txt
