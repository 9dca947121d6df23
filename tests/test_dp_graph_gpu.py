"""The data-parallel training path on one GPU: collectives forced on a one-rank RCCL group, eager and
captured into HIP graphs, against plain single-process training (see dp_graph_worker.py).  Runs in a
child process so that the process group never touches the other tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_dp_path_and_graph_capture_of_collectives():
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_graph_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-u", worker], capture_output=True, text=True, timeout=110, env=env)
    assert r.returncode == 0 and "WORKER_OK" in r.stdout, f"stdout:\n{r.stdout}\nstderr:\n{r.stderr[-3000:]}"
