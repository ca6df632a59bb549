"""The C-ABI library loads and exports every symbol include/igx.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

from isaacgym_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "igx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(igx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = _declared_symbols()
    assert "igx_ant_step" in names and "igx_last_error_string" in names
    for n in names:
        assert hasattr(lib, n), "libigx.so does not export %s" % n
        assert n in _lib.SYMBOLS, "isaacgym_amd._lib.SYMBOLS lacks %s" % n
    assert sorted(_lib.SYMBOLS) == names


def test_version_and_error_string():
    lib = _lib.lib()
    assert lib.igx_version() >= 120  # the header's version: the three newest entry points are in
    assert isinstance(lib.igx_last_error_string(), bytes)


def test_struct_sizes_match_header():
    # sizes computed from the header by the C compiler
    import subprocess
    import tempfile
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "igx.h"
int main(void){
  printf("CartpoleParams %zu\n", sizeof(igx_cartpole_params));
  printf("CartpoleBuffers %zu\n", sizeof(igx_cartpole_buffers));
  printf("LocoParams %zu\n", sizeof(igx_loco_params));
  printf("LocoBuffers %zu\n", sizeof(igx_loco_buffers));
  printf("AllegroParams %zu\n", sizeof(igx_allegro_params));
  printf("AllegroBuffers %zu\n", sizeof(igx_allegro_buffers));
  printf("ShadowParams %zu\n", sizeof(igx_shadow_params));
  printf("ShadowBuffers %zu\n", sizeof(igx_shadow_buffers));
  printf("off_shadow_seed %zu\n", offsetof(igx_shadow_params, seed));
  printf("MotionTables %zu\n", sizeof(igx_motion_tables));
  printf("MotionStateOut %zu\n", sizeof(igx_motion_state_out));
  printf("AmpParams %zu\n", sizeof(igx_amp_params));
  printf("AmpBuffers %zu\n", sizeof(igx_amp_buffers));
  printf("VecTaskTail %zu\n", sizeof(igx_vec_task_tail));
  printf("Noise %zu\n", sizeof(igx_noise));
  printf("ActionStage %zu\n", sizeof(igx_action_stage));
  printf("off_noise %zu\n", offsetof(igx_vec_task_tail, obs_noise));
  printf("off_mode %zu\n", offsetof(igx_action_stage, mode));
  printf("off_seed %zu\n", offsetof(igx_allegro_params, seed));
  printf("off_clip %zu\n", offsetof(igx_vec_task_tail, clip_obs));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode()
    for line in out.strip().splitlines():
        name, size = line.split()
        if name == "off_seed":
            assert _abi.AllegroParams.seed.offset == int(size)
        elif name == "off_shadow_seed":
            assert _abi.ShadowParams.seed.offset == int(size)
        elif name == "off_clip":
            assert _abi.VecTaskTail.clip_obs.offset == int(size)
        elif name == "off_noise":
            assert _abi.VecTaskTail.obs_noise.offset == int(size)
        elif name == "off_mode":
            assert _abi.ActionStage.mode.offset == int(size)
        else:
            assert ctypes.sizeof(getattr(_abi, name)) == int(size), name


def test_null_arguments_are_rejected_without_a_gpu():
    lib = _lib.lib()
    rc = lib.igx_ant_step(None, None, 16, None)
    assert rc == _abi.IGX_E_NULL
    assert b"NULL" in lib.igx_last_error_string()
    p = _abi.LocoParams(num_dof=8, num_sensor_floats=24)
    b = _abi.LocoBuffers()
    assert lib.igx_ant_step(ctypes.byref(p), ctypes.byref(b), -1, None) == _abi.IGX_E_SIZE
    assert lib.igx_ant_step(ctypes.byref(p), ctypes.byref(b), 16, None) == _abi.IGX_E_NULL
    assert lib.igx_humanoid_step(ctypes.byref(p), ctypes.byref(b), 16, None) == _abi.IGX_E_PARAM


def test_every_entry_point_rejects_null_structs():
    lib = _lib.lib()
    for name in ("igx_cartpole_step", "igx_humanoid_step", "igx_allegro_pre_physics", "igx_allegro_post_physics",
                 "igx_allegro_post_physics_shard", "igx_humanoid_amp_step", "igx_shadow_pre_physics",
                 "igx_shadow_post_physics", "igx_shadow_post_physics_shard"):
        assert getattr(lib, name)(None, None, 4, None) == _abi.IGX_E_NULL, name
    for name in ("igx_cartpole_reset", "igx_ant_reset", "igx_humanoid_reset"):
        assert getattr(lib, name)(None, None, None, 4, None) == _abi.IGX_E_NULL, name
    assert lib.igx_humanoid_amp_reset(None, None, None, 4, None, None) == _abi.IGX_E_NULL
    assert lib.igx_motion_state(None, None, None, 4, None, None) == _abi.IGX_E_NULL
    assert lib.igx_vec_task_tail_step(None, 4, None) == _abi.IGX_E_NULL
    assert lib.igx_clamp(None, -1.0, 1.0, None, 4, None) == _abi.IGX_E_NULL
    assert lib.igx_noise_apply(None, None, None, 4, None) == _abi.IGX_E_NULL
    assert lib.igx_action_stage_step(None, 4, None) == _abi.IGX_E_NULL
    stage = _abi.ActionStage(mode=7, num_actions=8, clip_actions=1.0)
    assert lib.igx_action_stage_step(ctypes.byref(stage), 4, None) == _abi.IGX_E_PARAM
    stage = _abi.ActionStage(mode=_abi.IGX_ACT_FIRST_DOF, num_actions=2, num_dof=2, clip_actions=1.0, actions=16, targets=16)
    assert lib.igx_action_stage_step(ctypes.byref(stage), 4, None) == _abi.IGX_E_PARAM
    assert lib.igx_quat_mul(None, None, None, 4, None) == _abi.IGX_E_NULL
    assert lib.igx_quat_apply(None, None, None, 4, None) == _abi.IGX_E_NULL
    assert lib.igx_tf_combine(None, None, None, None, None, None, 4, None) == _abi.IGX_E_NULL
    assert lib.igx_compute_rot(*([None] * 11), 4, None) == _abi.IGX_E_NULL
    assert lib.igx_compute_heading_and_up(*([None] * 5), 3, *([None] * 5), 4, None) == _abi.IGX_E_PARAM
    assert lib.igx_quat_axis(None, 5, None, 4, None) == _abi.IGX_E_PARAM
    assert lib.igx_extract_positions_from_rigid_body_states(None, 286, 33, None, 4, None) == _abi.IGX_E_PARAM
    assert lib.igx_extract_positions_from_rigid_body_states(None, 285, 22, None, 4, None) == _abi.IGX_E_SIZE
    assert lib.igx_extract_positions_from_rigid_body_states(None, 286, 22, None, 4, None) == _abi.IGX_E_NULL
    assert lib.igx_extract_positions_from_rigid_body_states(None, 286, 22, None, 0, None) == 0
    assert lib.igx_normalize(None, None, 4, 5, None) == _abi.IGX_E_PARAM
    assert lib.igx_torch_rand_float(0.0, 1.0, None, 0, 0, None, 4, None) == _abi.IGX_E_NULL
    assert lib.igx_get_euler_xyz(None, None, None, None, 0, None) == 0
    # empty batches are valid no-ops even with NULL pointers
    assert lib.igx_clamp(None, -1.0, 1.0, None, 0, None) == 0
    p = _abi.AmpParams(num_amp_obs_steps=1)
    b = _abi.AmpBuffers()
    assert lib.igx_humanoid_amp_step(ctypes.byref(p), ctypes.byref(b), 4, None) == _abi.IGX_E_PARAM


def test_binding_stub_in_the_integration_guide_matches_the_header():
    """The ctypes stub a maintainer of the reference would paste must describe today's structs."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# isaacgymenvs/utils/igx_binding\.py.*?)```", text, flags=re.S).group(1)
    block = block.replace("/path/to/libigx.so", _lib.LIB_PATH)
    ns = {}
    exec(compile(block, "igx_binding.py", "exec"), ns)
    for mine, theirs in ((_abi.LocoParams, ns["LocoParams"]), (_abi.LocoBuffers, ns["LocoBuffers"])):
        assert ctypes.sizeof(mine) == ctypes.sizeof(theirs)
        assert [f[0] for f in mine._fields_] == [f[0] for f in theirs._fields_]
        for f in mine._fields_:
            assert getattr(mine, f[0]).offset == getattr(theirs, f[0]).offset, f[0]
