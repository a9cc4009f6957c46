timeout 900 python -m pytest tests/test_dispatch.py -m gpu -q -x 2>&1 | tail -40
