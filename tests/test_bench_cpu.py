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
