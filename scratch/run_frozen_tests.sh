# tests of the dx-only backward paths (one GPU process, bounded)
timeout -k 10 300 python -u -m pytest tests/test_module_gpu.py tests/test_head_gpu.py tests/test_pair_gpu.py -k "frozen" -x -q --timeout 120 --timeout-method thread 2>&1 | tail -30
