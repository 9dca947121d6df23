"""CPU tests: the oracle against the golden vectors, its error behaviour, and its layout."""
import numpy as np
import pytest

import oracle
from helpers import (FWD_ATOL_SCALE, FWD_RTOL, GRAD_ATOL_SCALE, GRAD_RTOL, assert_close, expected,
                     gru_case_names, load_gru_case)


@pytest.mark.parametrize("name", gru_case_names())
@pytest.mark.parametrize("tag,dtype,tight", [("f32", np.float32, False), ("f64", np.float64, True)])
def test_oracle_matches_torch_cpu(name, tag, dtype, tight):
    case = load_gru_case(name)
    out = oracle.gru(case["x"], case["weights"], case["lengths"], case["dy"], case["dhn"], dtype=dtype)
    ref = expected(case, tag)
    # fp64 oracle vs fp64 torch agree to rounding; fp32 to the parity bar of the GPU tests
    fr, fa, gr, ga = (1e-12, 1e-13, 1e-11, 1e-12) if tight else (FWD_RTOL, FWD_ATOL_SCALE, GRAD_RTOL, GRAD_ATOL_SCALE)
    assert_close(out["y"], ref["y"], fr, fa, "y")
    assert_close(out["h_n"], ref["h_n"], fr, fa, "h_n")
    assert_close(out["dx"], ref["dx"], gr, ga, "dx")
    for l, (g, r) in enumerate(zip(out["grads"], ref["grads"])):
        for nm, a, b in zip(("dw_ih", "dw_hh", "db_ih", "db_hh"), g, r):
            assert_close(a, b, gr, ga, f"{nm}_l{l}")


def test_oracle_padded_rows_are_zero_and_h_n_is_last_valid_step():
    case = load_gru_case("gru_b3_t9_h24_l3_ragged")
    out = oracle.gru(case["x"], case["weights"], case["lengths"], case["dy"], case["dhn"])
    for b, n in enumerate(case["lengths"]):
        assert np.all(out["y"][b, n:] == 0)
        assert np.all(out["dx"][b, n:] == 0)
        np.testing.assert_array_equal(out["h_n"][-1, b], out["y"][b, n - 1])


def test_oracle_sequences_are_independent_of_batch_mates():
    case = load_gru_case("gru_b3_t9_h24_l3_ragged")
    full = oracle.gru(case["x"], case["weights"], case["lengths"])
    solo = oracle.gru(case["x"][1:2], case["weights"], case["lengths"][1:2])
    np.testing.assert_array_equal(full["y"][1], solo["y"][0])


def test_oracle_rejects_bad_lengths():
    case = load_gru_case("gru_b3_t9_h24_l3_ragged")
    with pytest.raises(RuntimeError, match="sorted in decreasing order"):
        oracle.gru(case["x"], case["weights"], [2, 5, 7])
    with pytest.raises(RuntimeError, match="greater than 0"):
        oracle.gru(case["x"], case["weights"], [7, 5, 0])


def test_philox_oracle_reproduces_the_published_known_answers():
    """Philox4x32-10 known-answer vectors of the Random123 library (kat_vectors: counter, key -> output), which
    pin the CPU restatement of the library's noise generator; then the float mapping: 24 bits, [0, 1)."""
    import oracle
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert [int(v) for v in oracle.philox4x32_10(ctr, key)] == want
    u = oracle.uniform(4, 0, 0)
    assert [float(v) for v in u] == [(w >> 8) / 2.0 ** 24 for w in kat[0][2]]
    big = oracle.uniform(100000, 7, 3)
    assert big.min() >= 0.0 and big.max() < 1.0 and abs(float(big.mean()) - 0.5) < 0.01
    # a shard of a stream is the slice of the whole stream, wherever it starts
    assert np.array_equal(oracle.uniform(1000, 7, 3, first=4321), oracle.uniform(6000, 7, 3)[4321:5321])
