# trainer / optimiser / head tests on the GPU (one process, bounded)
timeout -k 10 500 python -u -m pytest tests/test_timegan_host.py tests/test_full_size_gpu.py tests/test_head_gpu.py tests/test_data_scorer.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -15
