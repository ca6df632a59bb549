"""Re-run equality: every step entry point, run three times from identical inputs on a ragged environment count,
must leave byte-identical buffers.  (The reference has no race detector and neither has this pool; a kernel whose
result depended on the order in which waves or atomics land would show up here.)"""
import pytest
import torch

from isaacgym_amd import _abi, ops, synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = 100003   # not a multiple of any tile size


def _clone(d):
    return {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


def _same(a, b, what):
    for k in a:
        if isinstance(a[k], torch.Tensor) and k not in ("workspace", "qws"):  # scratch: candidates land in any order
            x, y = a[k].contiguous().view(torch.uint8), b[k].contiguous().view(torch.uint8)
            assert torch.equal(x, y), (what, k)


def _repeat(what, state, call, runs=3):
    outs = []
    for _ in range(runs):
        s = _clone(state)
        call(s)
        outs.append(s)
    torch.cuda.synchronize()
    for o in outs[1:]:
        _same(outs[0], o, what)
    return outs[0]


def test_cartpole_and_locomotion_steps_repeat_bit_for_bit():
    p = synthetic.cartpole_params(seed=3, rng_offset=4)
    _repeat("cartpole", synthetic.cartpole_state(N, 1, DEV), lambda s: ops.cartpole_step(p, s, N))
    for spec, fn in ((synthetic.ANT, ops.ant_step), (synthetic.HUMANOID, ops.humanoid_step)):
        p = synthetic.loco_params(spec, seed=3, rng_offset=4)
        st = synthetic.loco_state(spec, N, 2, DEV)
        st["reset"][::5] = 1   # in-kernel Philox resets
        _repeat("loco %d" % spec["num_dof"], st, lambda s: fn(p, s, N))


def test_hand_steps_repeat_bit_for_bit():
    for name, params, state, pre, post in (
            ("allegro", synthetic.allegro_params, synthetic.allegro_state, ops.allegro_pre_physics, ops.allegro_post_physics),
            ("shadow", synthetic.shadow_params, synthetic.shadow_state, ops.shadow_pre_physics, ops.shadow_post_physics)):
        p = params("full_state", seed=5, rng_offset=6)
        st = state(N, 3, DEV, "full_state")
        st["reset"][::4] = 1
        st["reset_goal"][::3] = 1

        def both(s):
            pre(p, s, N)
            post(p, s, N)
        out = _repeat(name, st, both)
        assert float(out["consecutive_successes"]) == float(out["consecutive_successes"])  # not NaN


def test_amp_and_imitation_steps_repeat_bit_for_bit():
    p = synthetic.amp_params(2)
    _repeat("amp", synthetic.amp_state(N, 4, DEV, 2), lambda s: ops.humanoid_amp_step(p, s, N))
    p = synthetic.him_params(seed=7, rng_offset=8)
    st = synthetic.him_state(N, 5, DEV)
    st["reset"][::6] = 1
    _repeat("humanoid_imitate", st, lambda s: ops.humanoid_imitate_step(p, s, N))

    # AntImitate: the fused step with the windowed select (atomics inside: the candidates arrive in any order)
    p = synthetic.loco_params(synthetic.ANT, seed=9, rng_offset=1)
    st = synthetic.loco_state(synthetic.ANT, N, 6, DEV)
    gen = torch.Generator(device=DEV).manual_seed(2)
    lo, hi = (torch.tensor(v, device=DEV, dtype=torch.float64) for v in synthetic.loco_limits(8))
    motion = lo + (hi - lo) * torch.rand(40, 8, generator=gen, device=DEV, dtype=torch.float64)
    st["target_pose_idx"] = torch.randint(0, 39, (N,), generator=gen, device=DEV)
    st["pose_diff"] = torch.zeros(N, dtype=torch.float64, device=DEV)
    st["qws"] = torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV)

    def imitate(s):
        bufs = {k: v for k, v in s.items() if k not in ("target_pose_idx", "pose_diff", "qws")}
        im = ops.imitate_args(motion, s["target_pose_idx"], s["pose_diff"], s["qws"], 0.99, 1.8, -2.0)
        ops.ant_imitate_step(p, bufs, im, N)
        s.update(bufs)
    outs = [_repeat("ant_imitate", st, imitate, runs=2) for _ in range(2)]
    out = outs[0]
    assert float(outs[0]["qws"].view(torch.float64)[5]) == float(outs[1]["qws"].view(torch.float64)[5])
    assert float(out["qws"].view(torch.float64)[5]) == float(torch.quantile(out["pose_diff"].cpu(), 0.99))


def test_selects_and_statistics_repeat_bit_for_bit():
    gen = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(N, generator=gen, device=DEV, dtype=torch.float64)
    got = set()
    for _ in range(3):
        ws = torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV)
        got.add((float(ops.quantile_f64(x, 0.37, ws)), float(ops.quantile_f64_windowed(x, 0.37, ws)),
                 float(ops.quantile_f64_windowed(x, 0.37, ws))))
    assert len(got) == 1 and len(set(got.pop())) == 1

    from isaacgym_amd.episode_stats import EpisodeStats
    rew = torch.rand(N, generator=gen, device=DEV)
    done = (torch.rand(N, generator=gen, device=DEV) < 0.1).to(torch.int64)
    totals = set()
    for _ in range(3):
        es = EpisodeStats(N, DEV)
        for _ in range(3):
            es.step(rew, done)
        totals.add(tuple(es.totals()))
    assert len(totals) == 1


def test_calls_on_two_streams_do_not_share_state():
    """The library is re-entrant: the same entry points running concurrently on two streams (own buffers, own
    workspaces) leave what the same calls leave one after the other."""
    n = 60001
    pa = synthetic.loco_params(synthetic.ANT, seed=3, rng_offset=4)
    ph = synthetic.allegro_params("full_state", seed=5, rng_offset=6)
    a0, h0 = synthetic.loco_state(synthetic.ANT, n, 2, DEV), synthetic.allegro_state(n, 3, DEV, "full_state")
    a0["reset"][::5] = 1
    h0["reset"][::4] = 1
    x = torch.randn(n, device=DEV, dtype=torch.float64)

    def work(a, h, ws):
        for _ in range(3):
            ops.ant_step(pa, a, n)
            ops.allegro_post_physics(ph, h, n)
            q = ops.quantile_f64_windowed(x, 0.9, ws)
        return q

    serial = [(_clone(a0), _clone(h0), torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV))
              for _ in range(2)]
    qs = [work(*s) for s in serial]
    torch.cuda.synchronize()
    conc = [(_clone(a0), _clone(h0), torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV))
            for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    qc = []
    for st, c in zip(streams, conc):
        st.wait_stream(torch.cuda.current_stream())
    for rep in range(1):
        for st, c in zip(streams, conc):
            with torch.cuda.stream(st):
                qc.append(work(*c))
    torch.cuda.synchronize()
    for s, c, q1, q2 in zip(serial, conc, qs, qc):
        _same(s[0], c[0], "ant on a side stream")
        _same(s[1], c[1], "allegro on a side stream")
        assert float(q1) == float(q2) == float(torch.quantile(x.cpu(), 0.9))
