python tools/kbench.py --nodevs --reps 7 2>&1 | tail -2
