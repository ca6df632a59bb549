"""AntImitate: CPU checker against the recording of the reference task class
(tests/golden/ant_imitate.npz), and the quantile restatement against torch.quantile."""
import numpy as np
import torch

from oracle import oracle
from tests import helpers as H


def test_ant_imitate_oracle_matches_reference():
    g = H.load_golden("ant_imitate")

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
