export TMPDIR=/tmp
mkdir -p gpurun_out/r02_zz
for c in 3 5; do timeout 900 python bench.py --config $c > gpurun_out/r02_zz/bench_c$c.json 2> gpurun_out/r02_zz/bench_c$c.err; python -c "
import json
d=json.load(open('gpurun_out/r02_zz/bench_c$c.json')); print('config $c:', d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d.get('parity'), d['breakdown_ms'], (d.get('latency_per_burst_ms') or {}).get('p50'), (d.get('latency_per_burst_ms') or {}).get('p99'))"; done
timeout 900 python tools/pmc_traffic.py 2>&1 | tail -3 | cut -c1-400
