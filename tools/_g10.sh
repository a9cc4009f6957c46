export TMPDIR=/tmp
python tools/pmc_traffic.py 2>&1 | tail -5
bash tools/gpu_round.sh r02_i 2>&1 | tail -40
