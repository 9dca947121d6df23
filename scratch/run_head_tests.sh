# head kernels against torch on the CPU (one GPU process, bounded)
timeout -k 10 300 python -u -m pytest tests/test_head_gpu.py tests/test_deferred_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -40
