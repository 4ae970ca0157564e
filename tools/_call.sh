python -m pytest tests/test_gpu_gp.py tests/test_gpu_gp_chunked.py tests/test_gpu_gp_lane.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | grep -E "^E  |^>|passed|failed" | head -12
