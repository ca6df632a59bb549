"""AntImitate: CPU checker against the recording of the reference task class
(tests/golden/ant_imitate.npz), and the quantile restatement against torch.quantile."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests import helpers as H


@pytest.mark.parametrize("recording", ["ant_imitate", "ant_imitate_extreme"])  # extreme: huge / tiny / subnormal / non-finite inputs
def test_ant_imitate_oracle_matches_reference(recording):
    g = H.load_golden(recording)

    def step(p, bufs, arrays, m):
        im = oracle.imitate_args(arrays, m["F"], m["quantile"], m["distance_floor"], m["death_cost"], m["N"])
        return oracle.ant_imitate_step(p, bufs, im, arrays, m["N"])

    H.ant_imitate_replay(g, step, to_buf=lambda a: np.ascontiguousarray(a).copy(), to_np=lambda a: a,
                         zeros=lambda shape, dt: np.zeros(shape, dt))


def test_quantile_restatement_equals_torch():
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 100, 101, 4097):
        x = rng.random(n) * 10
        x[: n // 3] = x[0]  # duplicates
        for q in (0.0, 0.5, 0.95, 0.99, 1.0):
            want = float(torch.quantile(torch.from_numpy(x), q))
            assert oracle.quantile_f64(x, q) == want, (n, q)


def test_radix_select_restatement_equals_torch_and_sort():
    """The shardable form of the quantile (per-pass histograms) against torch.quantile, with ties,
    negative values, signed zeros and the two end points."""
    rng = np.random.default_rng(4)
    for n in (1, 2, 3, 257, 40001):
        x = rng.standard_normal(n) ** 3
        if n > 100:
            x[: n // 3] = x[0]          # heavy ties
            x[n // 3] = -0.0
            x[n // 3 + 1] = 0.0
        for q in (0.0, 0.37, 0.5, 0.99, 1.0):
            ws = oracle.quantile_workspace()
            oracle.quantile_begin(ws, q, n)
            for p in range(6):
                oracle.quantile_histogram(x, ws, p)
                assert int(oracle.quantile_hist_view(ws, p)[:2048].sum()) >= 1
            got = oracle.quantile_finish(ws)
            assert got == float(torch.quantile(torch.from_numpy(x), q)), (n, q)
            assert got == oracle.quantile_f64(x, q)


def test_quantile_random_ranks_and_weights_equal_torch():
    """Random (n, q): torch's CPU lerp is one fused multiply-add onto the nearer order statistic; a plain
    multiply-add differs in the last bit for about one case in ten, which fixed q = 0.99 cases miss."""
    rng = np.random.default_rng(4)
    for _ in range(1500):
        n = int(rng.integers(1, 60))
        x = rng.standard_normal(n) ** 3
        q = float(rng.random())
        assert oracle.quantile_f64(x, q) == float(torch.quantile(torch.from_numpy(x), q)), (n, q)


def test_quantile_restatement_propagates_nan_like_torch():
    rng = np.random.default_rng(3)
    for nan in (np.nan, -np.nan):
        x = rng.standard_normal(1001)
        x[17] = nan
        assert np.isnan(float(torch.quantile(torch.from_numpy(x), 0.99)))
        assert np.isnan(oracle.quantile_f64(x, 0.99))
    # and the reference's max(1.8, quantile) then keeps 1.8 (Python's max returns its first argument unless the
    # second compares greater), so environments further than 1.8 are still reset
    assert max(1.8, torch.tensor(float("nan"))) == 1.8
