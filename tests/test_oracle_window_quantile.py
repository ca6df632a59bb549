"""The quantile in two collectives: the product's protocol (ops.WindowQuantile.run) over the CPU restatement of its
kernels, several shards inside one process, against torch.quantile over the concatenated shards - bit for bit."""
import numpy as np
import pytest
import torch

from isaacgym_amd import _abi
from tests.helpers import LocalShards, oracle_window_quantile


def _split(x, world, rng):
    cuts = np.sort(rng.integers(0, x.size + 1, world - 1))
    return [torch.from_numpy(np.ascontiguousarray(p)) for p in np.split(x, cuts)]


def _run(x, q, world, rng, capacity=4096, objs=None, reset=False):
    objs = objs or [oracle_window_quantile(capacity) for _ in range(world)]
    res = LocalShards(objs).run(_split(x, world, rng), q, reset=reset)
    vals = [float(o.result()) for o in objs]
    return objs, res, vals


CASES = {
    "normal": lambda r, n: r.standard_normal(n),
    "narrow": lambda r, n: 1.0 + 1e-9 * r.random(n),
    "wide": lambda r, n: np.exp(40.0 * r.standard_normal(n)) * r.choice([-1.0, 1.0], n),
    "ties": lambda r, n: r.integers(0, 5, n).astype(np.float64),
    "constant": lambda r, n: np.full(n, 2.5),
    "zeros_signed": lambda r, n: r.choice([0.0, -0.0, 1.0], n),
    "inf": lambda r, n: np.where(r.random(n) < 0.3, np.inf, r.standard_normal(n)),
    "denormal": lambda r, n: r.integers(0, 1000, n).astype(np.uint64).view(np.float64),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("world", [1, 2, 3])
def test_window_protocol_equals_torch(name, world):
    rng = np.random.default_rng(hash((name, world)) & 0xffff)
    for n in (1, 2, 7, 1000, 20000):
        x = CASES[name](rng, n)
        for q in (0.99, 0.5, 0.0, 1.0, float(rng.random())):
            objs, res, vals = _run(x, q, world, rng)
            want = float(torch.quantile(torch.from_numpy(x), q))
            assert all(res), (name, n, q)
            assert all(v == want or (np.isnan(v) and np.isnan(want)) for v in vals), (name, n, q, vals, want)
            assert len({o.workspace.numpy().tobytes() for o in objs}) == 1  # every rank holds the same state


def test_small_capacity_refines_and_resolves_heavy_ties():
    rng = np.random.default_rng(5)
    x = np.concatenate([np.full(5000, 3.0), rng.standard_normal(3000)])
    rng.shuffle(x)
    for q in (0.2, 0.5, 0.9, 0.375 + 1e-9):
        objs, res, vals = _run(x, q, 2, rng, capacity=16)
        assert all(res) and vals[0] == vals[1] == float(torch.quantile(torch.from_numpy(x), q)), q
    # the located bin is one single value held by more keys than the capacity
    assert objs[0].rounds >= 3


def test_nan_anywhere_gives_nan():
    rng = np.random.default_rng(6)
    x = rng.standard_normal(5000)
    x[1234] = np.nan
    objs, res, vals = _run(x, 0.99, 2, rng)
    assert all(res) and all(np.isnan(v) for v in vals)
    x[1234] = -np.nan
    objs, res, vals = _run(x, 0.5, 3, rng)
    assert all(res) and all(np.isnan(v) for v in vals)


def test_window_follows_drifting_data_in_two_collectives():
    """After the first call the window sits on the value: slowly drifting data costs one histogram and one gather."""
    rng = np.random.default_rng(7)
    objs = [oracle_window_quantile() for _ in range(2)]
    counts = []
    for step in range(30):
        x = np.abs(rng.standard_normal(200000)) * (1.0 + 0.01 * step) + 0.002 * step
        q = 0.5  # the median: on a cold window its bin is far too full, so the first call has to refine
        _, res, vals = _run(x, q, 2, rng, objs=objs)
        assert all(res) and vals[0] == vals[1] == float(torch.quantile(torch.from_numpy(x), q)), step
        counts.append(objs[0].collectives)
    assert 3 <= counts[0] <= 4 and all(c == 2 for c in counts[2:]), counts


def test_window_recovers_when_the_value_jumps():
    rng = np.random.default_rng(8)
    objs = [oracle_window_quantile() for _ in range(2)]
    for scale in (1.0, 1.0, 1e6, 1e-6, -1.0, 1.0, 0.0):
        x = rng.standard_normal(50000) * abs(scale) + (5.0 if scale < 0 else 0.0) * scale
        _, res, vals = _run(x, 0.99, 2, rng, objs=objs)
        assert all(res) and vals[0] == vals[1] == float(torch.quantile(torch.from_numpy(x), 0.99)), scale
        assert objs[0].collectives <= _abi.IGX_QUANTILE_WINDOW_MAX_ROUNDS + 1


def test_give_up_is_reported_and_forgets_the_window():
    """A histogram that does not hold n_total elements (a rank that skipped the all-reduce) cannot be located."""
    obj = oracle_window_quantile()
    x = torch.from_numpy(np.random.default_rng(9).standard_normal(1000))
    ok = obj.run(x, 0.5, 5000, lambda t: None, lambda t: t.unsqueeze(0))
    assert ok is False and obj.fallbacks == 1
    assert obj.run(x, 0.5, 1000, lambda t: None, lambda t: t.unsqueeze(0)) is True
    assert float(obj.result()) == float(torch.quantile(x, 0.5))


def test_debug_checks_catch_a_gather_that_does_not_add_up(monkeypatch):
    """With IGX_DEBUG_CHECKS the protocol reads the state back after the finish: slots that do not hold the located
    bin (here: a gather that returns only this rank's slots for both ranks' worth of keys) raise instead of
    leaving a NaN threshold behind."""
    from isaacgym_amd import _lib
    rng = np.random.default_rng(10)
    x = torch.from_numpy(rng.standard_normal(4000))
    obj = oracle_window_quantile()
    ok = obj.run(x[:2000].contiguous(), 0.5, 4000, lambda t: t.mul_(2), lambda t: t.unsqueeze(0))  # "two ranks", one gathered
    assert ok and np.isnan(float(obj.result()))
    monkeypatch.setattr(_lib, "DEBUG_CHECKS", True)
    obj = oracle_window_quantile()
    with pytest.raises(RuntimeError, match="do not hold the located bin"):
        obj.run(x[:2000].contiguous(), 0.5, 4000, lambda t: t.mul_(2), lambda t: t.unsqueeze(0))
    obj = oracle_window_quantile()
    assert obj.run(x, 0.5, 4000, lambda t: t, lambda t: t.unsqueeze(0))
    assert float(obj.result()) == float(torch.quantile(x, 0.5))
