"""Buffer sizes are validated on the host before any launch (isaacgym_amd._lib.check_sizes): a buffer
shorter than n_envs x width would be an out-of-bounds device access.  The host logic runs on CPU
tensors here; the GPU half goes through the real entry points."""
import pytest
import torch

from isaacgym_amd import _abi, _lib, ops, synthetic

N = 300


def _cases(dev):
    amp = synthetic.amp_state(N, 1, dev)
    return [
        ("loco", _abi.LocoBuffers, synthetic.loco_params(synthetic.ANT), synthetic.loco_state(synthetic.ANT, N, 1, dev),
         ops.ant_step, "obs"),
        ("humanoid", _abi.LocoBuffers, synthetic.loco_params(synthetic.HUMANOID),
         synthetic.loco_state(synthetic.HUMANOID, N, 1, dev), ops.humanoid_step, "dof_force"),
        ("allegro", _abi.AllegroBuffers, synthetic.allegro_params("full_state"),
         synthetic.allegro_state(N, 1, dev, "full_state", True), ops.allegro_post_physics, "states"),
        ("shadow", _abi.ShadowBuffers, synthetic.shadow_params("full_state"),
         synthetic.shadow_state(N, 1, dev, "full_state", True), ops.shadow_post_physics, "rigid_body_states"),
        ("amp", _abi.AmpBuffers, synthetic.amp_params(), amp, ops.humanoid_amp_step, "amp_obs"),
        ("humanoid_imitate", _abi.HumanoidImitateBuffers, synthetic.him_params(), synthetic.him_state(N, 1, dev),
         ops.humanoid_imitate_step, "rigid_body_states"),
    ]


def _short(t):
    return t.reshape(-1)[:t.numel() - 1].clone()


@pytest.mark.parametrize("idx", range(6))
def test_host_size_logic(idx):
    name, cls, p, t, _, field = _cases("cpu")[idx]
    _lib.check_sizes(cls, p, t, N, name)          # the synthetic state is exactly large enough
    _lib.check_sizes(cls, p, t, 0, name)
    for k in ("obs", field):
        with pytest.raises(ValueError, match=k):
            _lib.check_sizes(cls, p, dict(t, **{k: _short(t[k])}), N, name)
    with pytest.raises(ValueError, match="needs"):
        _lib.check_sizes(cls, p, t, N + 1, name)   # n_envs beyond what the buffers hold
    with pytest.raises(ValueError):
        _lib.check_sizes(cls, p, t, -1, name)
    if "workspace" in t:
        ws = t["workspace"]
        short = ws.reshape(-1)[:_abi.IGX_ALLEGRO_WORKSPACE_BYTES(N) // ws.element_size() - 1].clone()
        with pytest.raises(ValueError, match="workspace"):
            _lib.check_sizes(cls, p, dict(t, workspace=short), N, name)


def test_widths_follow_the_parameters():
    p = synthetic.allegro_params("full_no_vel")
    assert _abi.required_elements(_abi.AllegroBuffers, p, 10)["obs"] == (500, False)
    assert _abi.required_elements(_abi.AllegroBuffers, p, 10)["workspace"] == (32, True)
    p = synthetic.amp_params(num_amp_obs_steps=3)
    assert _abi.required_elements(_abi.AmpBuffers, p, 2)["amp_obs"] == (2 * 3 * 105, False)
    assert _abi.required_elements(_abi.LocoBuffers, synthetic.loco_params(synthetic.HUMANOID), 1)["obs"] == (108, False)
    assert _abi.required_elements(_abi.LocoBuffers, synthetic.loco_params(synthetic.ANT), 1)["obs"] == (60, False)
    assert _abi.required_elements(_abi.EpisodeBuffers, None, 256)["workspace"] == (96, True)


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(6))
def test_rejects_bad_arguments(idx):
    """Through the product entry points on the device: short obs, short workspace, n_envs > buffers."""
    name, cls, p, t, step, field = _cases("cuda:0")[idx]
    with pytest.raises(ValueError, match="obs"):
        step(p, dict(t, obs=_short(t["obs"])), N)
    with pytest.raises(ValueError, match=field):
        step(p, dict(t, **{field: _short(t[field])}), N)
    with pytest.raises(ValueError, match="needs"):
        step(p, t, N + 64)
    if "workspace" in t:
        with pytest.raises(ValueError, match="workspace"):
            step(p, dict(t, workspace=t["workspace"].reshape(-1)[:8].clone()), N)
    before = {k: v.clone() for k, v in t.items() if v is not None}
    step(p, t, N)  # and the well-formed call still runs
    torch.cuda.synchronize()
    assert not torch.equal(before["obs"], t["obs"])


@pytest.mark.gpu
def test_env_ids_range_with_debug_checks(monkeypatch):
    monkeypatch.setattr(_lib, "DEBUG_CHECKS", True)
    p = synthetic.loco_params(synthetic.ANT)
    t = synthetic.loco_state(synthetic.ANT, N, 1, "cuda:0")
    ops.ant_reset(p, t, torch.tensor([0, N - 1], device="cuda:0"))
    with pytest.raises(ValueError, match="env_ids"):
        ops.ant_reset(p, t, torch.tensor([0, N], device="cuda:0"))
    a = synthetic.allegro_state(N, 1, "cuda:0", "full_state", False)
    a["object_indices"] = a["object_indices"] + 3 * N
    with pytest.raises(ValueError, match="object_indices"):
        ops.allegro_post_physics(synthetic.allegro_params("full_state"), a, N)
