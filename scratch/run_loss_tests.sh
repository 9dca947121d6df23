# loss kernels against the torch composition (one GPU process, bounded)
timeout -k 10 300 python -u -m pytest tests/test_loss_gpu.py tests/test_timegan_host.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -8
