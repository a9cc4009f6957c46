#!/bin/bash
# The closing GPU visit of round 4: everything profiles/r04_* quotes, from one build.  tools/gpu_r4_final.sh [tag]
# R433_SKIP_WAVE_PMC=1 leaves the k_wave counter passes out (stream_kernels.hip unchanged since they were taken).
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export R433_PMC_TAG=$TAG
echo "== host"; { nproc; cat /sys/fs/cgroup/cpu.max; } 2>&1 | tee $OUT/host.txt
echo "== pytest -m gpu"
( time timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -12 ) 2>&1 | grep -v amdgpu.ids | tee $OUT/pytest.txt
echo "== bench (default command)"
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3
python tools/jq.py value ms_per_step roofline breakdown_ms hbm_resident parity d2h_bytes_per_step_per_gpu < $OUT/bench.json
echo "== rocprofv3 kernel trace of the resident pipeline"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --quick --resident --exclusive 3 > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err )
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt | head -30
find $OUT/prof -name '*.db' -size +20M -delete
echo "== kbench: the detection kernel alone, by grid size; lazy tiles on / off; pair / single"
{ for n in 1024 2048 4096 8192; do for f in 0 262144 4096; do timeout 300 python tools/kbench.py --nodevs --reps 6 --streams $n --debug $f 2>&1 | tail -1; done; done
  timeout 300 python tools/kbench.py --reps 4 --streams 8192 2>&1 | tail -1
  timeout 300 python tools/kbench.py --reps 4 --streams 8192 --debug 131072 2>&1 | tail -1
  timeout 300 python tools/kbench.py --nodevs --cs16 2>&1 | tail -1
  timeout 300 python tools/kbench.py --nodevs --fsk-cu8 2>&1 | tail -1
  timeout 300 python tools/lazy_stats.py 1024 2>&1 | tail -2; } 2>&1 | grep -v amdgpu.ids | tee $OUT/kbench.txt
echo "== PMC: traffic (FETCH_SIZE / WRITE_SIZE), issue (SQ counters, whole kernel and producers alone)"
if [ -z "$R433_SKIP_WAVE_PMC" ]; then
timeout 1200 python tools/pmc_traffic.py 2>&1 | tail -5 | cut -c1-300
timeout 900 python tools/pmc_issue.py 2>&1 | tail -32
fi
echo "== PMC: slicers"
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  bash tools/pmc_run.sh ${TAG}_slice_$(echo $pmc | cut -d' ' -f2) "$pmc" --streams 8192 2>&1 | grep k_slice | cut -c1-130
done | tee $OUT/pmc_slice.txt
echo "== configs 3 / 4 / 5 at full size"
for c in 3 4 5; do timeout 900 python bench.py --config $c > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; python tools/jq.py value ms_per_step roofline parity breakdown_ms < $OUT/bench_c$c.json | cut -c1-900; done
echo "== CLI drop-in"
timeout 600 tools/cli_bench.sh 1024 $OUT 2>&1 | tail -3
echo "== the C pipeline host (one GPU; -g 0 = every visible GPU)"
{ timeout 300 bash tools/pipeline_host_bench.sh 8192 2>&1 | tail -9
  cd /tmp/cli_bench && for rep in 1 2; do $GRAFT_REPO_ROOT/dropin/_build/pipeline_host_hip -q -g 0 -b 1024 -p $(ls s*_433.92M_250k.cu8 | head -8192 | tr '\n' ' ') 2>&1 | tail -2 | sed 's/^/[-g 0 -b 1024 -p] /'; done; cd $GRAFT_REPO_ROOT; } 2>&1 | tee $OUT/pipeline_host.txt
echo "== timeline of the pipeline (real decoders, stateless flags, 24 threads)"
timeout 300 python tools/leg_timeline.py 12 3 2 24 1 2>&1 | grep -v amdgpu.ids | tee $OUT/leg_timeline.txt | head -12
echo "== where the CPU quota goes"
timeout 200 python tools/cpu_budget.py 12 2>&1 | grep -v amdgpu.ids | tee $OUT/cpu_budget.txt
echo "== dispatch trace"
timeout 200 python tools/dispatch_trace.py 24 1 2>&1 | grep "r.dispatch\|replay" | cut -c1-400 | tee $OUT/dispatch_trace.txt
echo "== probes"
{ timeout 120 python tools/pcie_probe.py; } 2>&1 | grep -v amdgpu.ids | tee $OUT/probes.txt
echo "== fuzz (GPU, 6000 cases)"
timeout 1500 python tools/fuzz_emu.py --gpu ${R433_FUZZ_CASES:-6000} 500000 2>&1 | tail -1 | tee $OUT/fuzz.txt
grep thrott /sys/fs/cgroup/cpu.stat | tee -a $OUT/host.txt
ls $OUT
