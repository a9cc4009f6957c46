timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r02_h/pytest2.txt
