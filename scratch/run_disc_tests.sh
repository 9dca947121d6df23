# parameter updates with and without the gradients a sub-step discards (one GPU process, bounded)
timeout -k 10 300 python -u -m pytest tests/test_timegan_host.py -k "discarded" -x -q --timeout 120 --timeout-method thread 2>&1 | tail -30
