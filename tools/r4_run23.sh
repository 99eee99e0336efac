timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "pipelined" 2>&1 | tail -8
