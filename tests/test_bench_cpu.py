"""bench.py bookkeeping that does not need a GPU."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_work_matches_the_stated_formulas():
    b = _bench()
    # one sequence through one L-layer stack, forward+backward, In == H:
    # FLOPs = 36 L T H^2, bytes = 56 L T H (+ 12 L (6H^2 + 6H) of weights per launch)
    for (B, T, H, L) in [(128, 24, 24, 3), (1024, 24, 96, 3), (2048, 256, 512, 3)]:
        flops, bytes_ = b.algorithmic_work(B, T, H, H, L)
        assert flops == 36.0 * L * T * H * H * B
        assert bytes_ == 56.0 * L * T * H * B + 12.0 * L * (6 * H * H + 6 * H)
    flops, bytes_ = b.algorithmic_work(128, 24, 24, 24, 3)
    assert abs(flops - 191.1e6) < 0.1e6 and abs(bytes_ - 12.5e6) < 0.1e6      # C2 totals


def test_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (before touching any GPU),
    they form one process group, and exactly one result line comes out, from rank 0, with n_gpus == 2."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["process_group_ranks"] == 2


def test_a_failing_rank_fails_the_launch():
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "no-such-backend",
                        "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
