"""MotionLib: AMP reference-motion library with device-resident frame tables.

Mirror of the reference's tasks/amp/utils_amp/motion_lib.py (class MotionLib):
same constructor, `sample_motions`, `sample_time`, `get_motion_length`,
`get_motion_state` and `_calc_frame_blend`.  What differs is where the work
happens: the per-frame tables of every motion (forward kinematics, joint
velocities) are built once at load time and kept on the device, and
`get_motion_state` is one HIP kernel (fp64 frame index / blend, slerp, local
rotation -> dof) instead of a Python loop of numpy gathers followed by eleven
host-to-device copies.  `get_amp_obs` additionally fuses
build_amp_observations for the imitation-sampling path.

Sampling (`sample_motions`, `sample_time`) stays on the host numpy RNG exactly
as in the reference, so a seeded run draws the same (ids, times).

Motion files: `.npz` with plain arrays (rotation [F,J,4] xyzw, root_translation
[F,3], global_velocity [F,J,3], global_angular_velocity [F,J,3], parent_indices
[J], local_translation [J,3], fps, is_local), the reference's own `.npy` (a
pickled dict written by poselib's SkeletonMotion.to_file; read through an
unpickler that resolves numpy arrays, dtypes, scalars and OrderedDict and
nothing else - never through np.load(allow_pickle=True)), or a `.yaml` motion set
(`motions: [{file, weight}, ...]`, paths relative to the yaml).
"""
import ctypes as C
import os
import pickle

import numpy as np
import torch
import yaml

from .... import _abi, _lib, ops

# tasks/amp/humanoid_amp_base.py:43-44
DOF_BODY_IDS = [1, 2, 3, 4, 6, 7, 9, 10, 11, 12, 13, 14]
DOF_OFFSETS = [0, 3, 6, 9, 10, 13, 14, 17, 18, 21, 24, 25, 28]


# ---- poselib/core/rotation3d.py arithmetic (load time only, torch CPU) ------
def _quat_mul(a, b):  # rotation3d.py:36-49 (16-multiply form)
    x1, y1, z1, w1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    x2, y2, z2, w2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    y = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2
    z = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2
    return torch.stack([x, y, z, w], dim=-1)


def _quat_normalize(q):  # rotation3d.py:52-60, :73-79, :116-121
    z = (q[..., 3:] < 0).float()
    q = (1 - 2 * z) * q
    return q / q.norm(p=2, dim=-1).unsqueeze(-1).clamp(min=1e-9)


def _quat_conjugate(x):
    return torch.cat([-x[..., :3], x[..., 3:]], dim=-1)


def _quat_mul_norm(x, y):  # rotation3d.py:220-226
    return _quat_normalize(_quat_mul(x, y))


def _quat_rotate(rot, vec):  # rotation3d.py:229-235
    other_q = torch.cat([vec, torch.zeros_like(vec[..., :1])], dim=-1)
    return _quat_mul(_quat_mul(rot, other_q), _quat_conjugate(rot))[..., :3]


def _quat_angle_axis(x):  # rotation3d.py:254-264
    s = 2 * (x[..., 3] ** 2) - 1
    angle = s.clamp(-1, 1).arccos()
    axis = x[..., :3]
    axis = axis / axis.norm(p=2, dim=-1, keepdim=True).clamp(min=1e-9)
    return angle, axis


class _Motion:
    """The fields of a poselib SkeletonMotion that the library reads."""

    def __init__(self, rotation, root_translation, global_velocity, global_angular_velocity, parent_indices,
                 local_translation, fps, is_local=True):
        f = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
        self.rotation = f(rotation)
        self.root_translation = f(root_translation)
        self.global_velocity = f(global_velocity)
        self.global_angular_velocity = f(global_angular_velocity)
        self.parent_indices = [int(p) for p in np.asarray(parent_indices).tolist()]
        self.skel_local_translation = f(local_translation)
        self.fps = fps
        self.is_local = bool(is_local)
        self.num_frames = self.rotation.shape[0]
        self.num_joints = self.rotation.shape[1]
        self._forward_kinematics()

    def _forward_kinematics(self):
        """skeleton3d.py:431-452 (global_transformation) and :490-510 (local_rotation)."""
        F, J = self.num_frames, self.num_joints
        if self.is_local:
            local_rot = self.rotation
        else:
            local_rot = torch.zeros_like(self.rotation)
            for j, p in enumerate(self.parent_indices):
                if p == -1:
                    local_rot[:, j] = self.rotation[:, j]
                else:
                    local_rot[:, j] = _quat_mul_norm(_quat_conjugate(self.rotation[:, p]), self.rotation[:, j])
        local_t = self.skel_local_translation.unsqueeze(0).expand(F, J, 3).clone()
        local_t[:, 0] = self.root_translation
        g_rot, g_t = [], []
        for j, p in enumerate(self.parent_indices):
            if p == -1:
                g_rot.append(local_rot[:, j])
                g_t.append(local_t[:, j])
            else:  # rotation3d.py:346-356 transform_mul
                g_rot.append(_quat_mul_norm(g_rot[p], local_rot[:, j]))
                g_t.append(_quat_rotate(g_rot[p], local_t[:, j]) + g_t[p])
        self.local_rotation = local_rot
        self.global_rotation = torch.stack(g_rot, dim=1) if self.is_local else self.rotation
        self.global_translation = torch.stack(g_t, dim=1)
        self.global_root_velocity = self.global_velocity[:, 0]
        self.global_root_angular_velocity = self.global_angular_velocity[:, 0]


_NPY_GLOBALS = {  # everything a poselib motion file needs to be rebuilt: arrays, dtypes, scalars, ordered dicts
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"), ("collections", "OrderedDict"),
}


class _RestrictedUnpickler(pickle.Unpickler):
    """Unpickler that resolves only the globals in _NPY_GLOBALS: a file naming anything else (which is how a
    pickle makes the loader call arbitrary code) is rejected before that object is looked up."""

    def find_class(self, module, name):
        if (module, name) not in _NPY_GLOBALS:
            raise pickle.UnpicklingError("motion file refers to %s.%s, which a poselib motion does not need; "
                                         "refusing to load it" % (module, name))
        return super().find_class(module, name)


def read_poselib_npy(path):
    """The dict inside a poselib .npy (np.save of a Python dict = an object array holding one pickled dict,
    poselib/core/backend/abstract.py:116-150), read without np.load(allow_pickle=True): the .npy header is
    parsed here and the payload goes through _RestrictedUnpickler."""
    with open(path, "rb") as fh:
        version = np.lib.format.read_magic(fh)
        if version == (1, 0):
            shape, _, dtype = np.lib.format.read_array_header_1_0(fh)
        elif version == (2, 0):
            shape, _, dtype = np.lib.format.read_array_header_2_0(fh)
        else:
            raise ValueError("%s: unsupported .npy version %r" % (path, version))
        if not dtype.hasobject or shape != ():
            raise ValueError("%s is not a poselib motion (expected one pickled dict, found %s array of shape %s)"
                             % (path, dtype, shape))
        obj = _RestrictedUnpickler(fh).load()
    d = obj.item() if isinstance(obj, np.ndarray) else obj
    if not isinstance(d, dict):
        raise ValueError("%s does not hold a dict" % path)
    return d


def load_motion_file(path, allow_pickle=False):
    """.npz (plain arrays) or a poselib .npy.  `allow_pickle` is accepted for older callers and ignored:
    a .npy is always read through the restricted unpickler, never through numpy's pickle loader."""
    ext = os.path.splitext(path)[1]
    if ext == ".npz":
        z = np.load(path, allow_pickle=False)
        return _Motion(z["rotation"], z["root_translation"], z["global_velocity"], z["global_angular_velocity"],
                       z["parent_indices"], z["local_translation"], float(z["fps"]),
                       bool(z["is_local"]) if "is_local" in z.files else True)
    if ext == ".npy":
        d = read_poselib_npy(path)
        tree = d["skeleton_tree"]
        return _Motion(d["rotation"]["arr"], d["root_translation"]["arr"], d["global_velocity"]["arr"],
                       d["global_angular_velocity"]["arr"], tree["parent_indices"]["arr"],
                       tree["local_translation"]["arr"], d["fps"], d["is_local"])
    raise ValueError("unsupported motion file %r" % path)


def save_motion_npz(path, m):
    np.savez_compressed(path, rotation=m.rotation.numpy(), root_translation=m.root_translation.numpy(),
                        global_velocity=m.global_velocity.numpy(),
                        global_angular_velocity=m.global_angular_velocity.numpy(),
                        parent_indices=np.asarray(m.parent_indices, np.int64),
                        local_translation=m.skel_local_translation.numpy(), fps=np.float64(m.fps),
                        is_local=np.bool_(m.is_local))


def convert_npy(src, dst):
    """Convert a poselib .npy motion to the plain-array .npz format."""
    save_motion_npz(dst, load_motion_file(src))


class MotionLib:
    def __init__(self, motion_file, num_dofs, key_body_ids, device, allow_pickle=False):
        if num_dofs != _abi.IGX_AMP_NUM_DOF:
            raise ValueError("the AMP humanoid has %d dofs" % _abi.IGX_AMP_NUM_DOF)
        self._num_dof = num_dofs
        self._key_body_ids = np.asarray(key_body_ids)
        if self._key_body_ids.shape[0] != _abi.IGX_AMP_NUM_KEY_BODIES:
            raise ValueError("expected %d key bodies" % _abi.IGX_AMP_NUM_KEY_BODIES)
        self._device = torch.device(device)
        self._allow_pickle = allow_pickle
        self._load_motions(motion_file)
        self.motion_ids = torch.arange(len(self._motions), dtype=torch.long, device=self._device)

    # ---- reference API ----------------------------------------------------
    def num_motions(self):
        return len(self._motions)

    def get_total_length(self):
        return sum(self._motion_lengths)

    def get_motion(self, motion_id):
        return self._motions[motion_id]

    def sample_motions(self, n):  # motion_lib.py:61-65
        m = self.num_motions()
        return np.random.choice(m, size=n, replace=True, p=self._motion_weights)

    def sampling_cdf(self):
        """The cumulative weights np.random.choice searches (cumsum divided by its last entry), fp64, on the device."""
        if getattr(self, "_cdf", None) is None:
            cdf = self._motion_weights.cumsum()
            cdf /= cdf[-1]
            self._cdf = torch.from_numpy(np.ascontiguousarray(cdf, np.float64)).to(self._device)
        return self._cdf

    def sample_time(self, motion_ids, truncate_time=None):  # motion_lib.py:67-78
        phase = np.random.uniform(low=0.0, high=1.0, size=motion_ids.shape)
        motion_len = self._motion_lengths[motion_ids]
        if truncate_time is not None:
            assert truncate_time >= 0.0
            motion_len -= truncate_time
        return phase * motion_len

    def get_motion_length(self, motion_ids):
        return self._motion_lengths[motion_ids]

    def _calc_frame_blend(self, time, len, num_frames, dt):  # motion_lib.py:233-241 (host copy, fp64)
        phase = np.clip(time / len, 0.0, 1.0)
        frame_idx0 = (phase * (num_frames - 1)).astype(int)
        frame_idx1 = np.minimum(frame_idx0 + 1, num_frames - 1)
        blend = (time - frame_idx0 * dt) / dt
        return frame_idx0, frame_idx1, blend

    def get_motion_state(self, motion_ids, motion_times, return_frames=False):  # motion_lib.py:83-153
        ids, times = self._to_device(motion_ids, motion_times)
        n = ids.shape[0]
        dev = self._device
        f32 = dict(dtype=torch.float32, device=dev)
        out = dict(root_pos=torch.empty(n, 3, **f32), root_rot=torch.empty(n, 4, **f32),
                   dof_pos=torch.empty(n, 28, **f32), root_vel=torch.empty(n, 3, **f32),
                   root_ang_vel=torch.empty(n, 3, **f32), dof_vel=torch.empty(n, 28, **f32),
                   key_pos=torch.empty(n, 4, 3, **f32))
        if return_frames:
            out.update(frame_idx0=torch.empty(n, dtype=torch.int64, device=dev),
                       frame_idx1=torch.empty(n, dtype=torch.int64, device=dev), blend=torch.empty(n, **f32))
        o, _ = _lib.make_buffers(_abi.MotionStateOut, out)
        with torch.cuda.device(dev):
            rc = _lib.lib().igx_motion_state(C.byref(self._tables), _lib.tensor_ptr(ids, torch.int64),
                                             _lib.tensor_ptr(times, torch.float64), C.c_int64(n), C.byref(o),
                                             _lib.current_stream(dev))
        _lib.check(rc)
        res = (out["root_pos"], out["root_rot"], out["dof_pos"], out["root_vel"], out["root_ang_vel"],
               out["dof_vel"], out["key_pos"])
        if return_frames:
            return res + (out["frame_idx0"], out["frame_idx1"], out["blend"])
        return res

    # ---- fused imitation-sampling path -------------------------------------
    def get_amp_obs(self, motion_ids, motion_times, local_root_obs, out=None, out_rows=None, out_row_stride=None,
                    out_col_offset=0):
        """build_amp_observations(get_motion_state(ids, times)) in one kernel.

        Without `out` returns a new [n,105] tensor; with `out` sample i is written to
        out.view(-1)[(out_rows[i] if out_rows is not None else i) * out_row_stride + out_col_offset:][:105]."""
        ids, times = self._to_device(motion_ids, motion_times)
        n = ids.shape[0]
        dev = self._device
        if out is None:
            out = torch.empty(n, _abi.IGX_AMP_OBS, dtype=torch.float32, device=dev)
            out_row_stride = _abi.IGX_AMP_OBS
        elif out_row_stride is None:
            raise ValueError("out_row_stride is required with out")
        # the converted copy (not only the caller's tensor) must outlive the asynchronous launch
        rows_t = None if out_rows is None else out_rows.to(dev, torch.int64).contiguous()
        if rows_t is not None and rows_t.shape[0] != n:
            raise ValueError("out_rows must name one row per sample")
        rows = None if rows_t is None else _lib.tensor_ptr(rows_t, torch.int64)
        with torch.cuda.device(dev):
            rc = _lib.lib().igx_motion_amp_obs(C.byref(self._tables), _lib.tensor_ptr(ids, torch.int64),
                                               _lib.tensor_ptr(times, torch.float64), C.c_int64(n),
                                               C.c_int32(int(bool(local_root_obs))), _lib.tensor_ptr(out, torch.float32),
                                               rows, C.c_int64(out_row_stride), C.c_int64(out_col_offset),
                                               _lib.current_stream(dev))
        _lib.check(rc)
        self._last_rows = rows_t  # released when the next call replaces it: by then this launch has been enqueued
        return out

    def get_amp_obs_steps(self, motion_ids, motion_times, num_steps, step_dt, local_root_obs, out):
        """The batch of fetch_amp_obs_demo (humanoid_amp.py:146-171) for drawn (ids, times0): out[i, k] = observation
        of motion ids[i] at times0[i] + (-step_dt * k).  One launch; the host sends n ids and times, the tiling over
        the history steps happens in the kernel (same fp64 operations)."""
        ids, times = self._to_device(motion_ids, motion_times)
        ops.motion_amp_obs_steps(self._tables, ids, times, num_steps, step_dt, local_root_obs, out)
        return out

    def sample_on_device(self, n, seed, rng_offset, random_time=True, out=None):
        """sample_motions + sample_time for n samples without the host: (ids int64 [n], times float64 [n]) on the
        device, drawn from the library's Philox stream keyed by (seed, rng_offset, sample index)."""
        dev = self._device
        if out is None:
            out = (torch.empty(n, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.float64, device=dev))
        ops.motion_sample(self._tables, self.sampling_cdf(), out[0], out[1], random_time=random_time, seed=seed,
                          rng_offset=rng_offset)
        return out

    # ---- loading ------------------------------------------------------------
    def _to_device(self, motion_ids, motion_times):
        if not torch.is_tensor(motion_ids) and not torch.is_tensor(motion_times):  # the reference's case: numpy draws
            ids = np.ascontiguousarray(motion_ids, dtype=np.int64)
            times = np.ascontiguousarray(motion_times, dtype=np.float64)
            if ids.shape != times.shape or ids.ndim != 1:
                raise ValueError("motion_ids and motion_times must be 1-D and of equal length")
            if ids.size and (int(ids.min()) < 0 or int(ids.max()) >= self.num_motions()):
                raise IndexError("motion id out of range")
            return self._stage(ids, times)
        ids = torch.as_tensor(np.ascontiguousarray(motion_ids, dtype=np.int64)) if not torch.is_tensor(motion_ids) \
            else motion_ids.to(torch.int64)
        times = torch.as_tensor(np.ascontiguousarray(motion_times, dtype=np.float64)) \
            if not torch.is_tensor(motion_times) else motion_times.to(torch.float64)
        if ids.shape != times.shape or ids.dim() != 1:
            raise ValueError("motion_ids and motion_times must be 1-D and of equal length")
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.num_motions()):
            raise IndexError("motion id out of range")
        return ids.to(self._device).contiguous(), times.to(self._device).contiguous()

    _STAGE_SLOTS = 4

    def _stage(self, ids, times):
        """Host draws to the device through pinned memory: ids and times side by side in one slot of a small ring,
        one asynchronous copy (two blocking copies out of pageable memory cost 28 of a fetch's 59 us).  The device
        tensors returned are views of the slot: good for the launches the caller enqueues next, overwritten (in
        stream order) once the ring has come round."""
        n = int(ids.shape[0])
        dev = self._device
        if torch.device(dev).type != "cuda":  # no CPU path: the kernels behind every query take device memory
            raise ValueError("libigx works on device tensors only (this MotionLib lives on %s)" % (dev,))
        if n == 0:
            return torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.float64, device=dev)
        ring = self.__dict__.get("_stage_ring")
        if ring is None or ring["cap"] < n:
            cap = max(1024, 1 << (n - 1).bit_length())
            slots = []
            for _ in range(self._STAGE_SLOTS):
                host = torch.empty(2 * cap, dtype=torch.int64, pin_memory=True)
                slots.append(dict(host=host, view=host.numpy(), dev=torch.empty(2 * cap, dtype=torch.int64, device=dev),
                                  event=None))
            ring = self._stage_ring = dict(cap=cap, next=0, slots=slots)
        slot = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % self._STAGE_SLOTS
        if slot["event"] is not None:
            slot["event"].synchronize()  # the copy that last read this slot's host memory (four fetches ago)
        slot["view"][:n] = ids
        slot["view"][n:2 * n].view(np.float64)[:] = times
        with torch.cuda.device(dev):
            slot["dev"][:2 * n].copy_(slot["host"][:2 * n], non_blocking=True)
            if slot["event"] is None:
                slot["event"] = torch.cuda.Event()
            slot["event"].record()
        return slot["dev"][:n], slot["dev"][n:2 * n].view(torch.float64)

    def _fetch_motion_files(self, motion_file):  # motion_lib.py:208-231
        ext = os.path.splitext(motion_file)[1]
        if ext == ".yaml":
            dir_name = os.path.dirname(motion_file)
            with open(os.path.join(os.getcwd(), motion_file), "r") as f:
                motion_config = yaml.load(f, Loader=yaml.SafeLoader)
            files, weights = [], []
            for entry in motion_config["motions"]:
                assert entry["weight"] >= 0
                files.append(os.path.join(dir_name, entry["file"]))
                weights.append(entry["weight"])
            return files, weights
        return [motion_file], [1.0]

    def _load_motions(self, motion_file):  # motion_lib.py:155-206
        self._motions, self._motion_files = [], []
        lengths, weights, fpss, dts, nframes = [], [], [], [], []
        files, file_weights = self._fetch_motion_files(motion_file)
        for f, w in zip(files, file_weights):
            m = load_motion_file(f, self._allow_pickle) if isinstance(f, str) else f
            if m.num_joints != _abi.IGX_AMP_NUM_BODIES:
                raise ValueError("%s: expected %d bodies, got %d" % (f, _abi.IGX_AMP_NUM_BODIES, m.num_joints))
            fps = m.fps
            dts.append(1.0 / fps)
            nframes.append(m.num_frames)
            lengths.append(1.0 / fps * (m.num_frames - 1))
            fpss.append(fps)
            m.dof_vels = self._compute_motion_dof_vels(m)
            self._motions.append(m)
            weights.append(w)
            self._motion_files.append(f)
        self._motion_lengths = np.array(lengths)
        self._motion_weights = np.array(weights, dtype=np.float64)
        self._motion_weights /= np.sum(self._motion_weights)
        self._motion_fps = np.array(fpss)
        self._motion_dt = np.array(dts)
        self._motion_num_frames = np.array(nframes)
        self._build_tables()

    def _compute_motion_dof_vels(self, motion):  # motion_lib.py:248-263, :295-324
        dt = 1.0 / motion.fps
        lr = motion.local_rotation
        diff = _quat_mul_norm(_quat_conjugate(lr[:-1]), lr[1:])
        angle, axis = _quat_angle_axis(diff)
        local_vel = (axis * angle.unsqueeze(-1) / dt).numpy()  # [F-1, J, 3]
        dof_vel = np.zeros([local_vel.shape[0], self._num_dof])
        for j, body in enumerate(DOF_BODY_IDS):
            off, size = DOF_OFFSETS[j], DOF_OFFSETS[j + 1] - DOF_OFFSETS[j]
            if size == 3:
                dof_vel[:, off:off + 3] = local_vel[:, body]
            else:
                dof_vel[:, off] = local_vel[:, body, 1]  # joint assumed to turn about y
        return np.concatenate([dof_vel, dof_vel[-1:]], axis=0)

    def _build_tables(self):
        dev = self._device
        cat = lambda xs: torch.cat(xs, dim=0).to(dev).contiguous()  # noqa: E731
        ms = self._motions
        kb = torch.as_tensor(self._key_body_ids, dtype=torch.long)
        offs = np.concatenate([[0], np.cumsum(self._motion_num_frames)[:-1]]).astype(np.int64)
        self._tabs = dict(
            motion_length=torch.as_tensor(self._motion_lengths, dtype=torch.float64).to(dev),
            motion_dt=torch.as_tensor(self._motion_dt, dtype=torch.float64).to(dev),
            motion_num_frames=torch.as_tensor(self._motion_num_frames, dtype=torch.int64).to(dev),
            frame_offset=torch.as_tensor(offs).to(dev),
            root_pos=cat([m.global_translation[:, 0] for m in ms]),
            root_rot=cat([m.global_rotation[:, 0] for m in ms]),
            local_rot=cat([m.local_rotation for m in ms]),
            root_vel=cat([m.global_root_velocity for m in ms]),
            root_ang_vel=cat([m.global_root_angular_velocity for m in ms]),
            key_pos=cat([m.global_translation[:, kb] for m in ms]),
            dof_vel=cat([torch.as_tensor(m.dof_vels, dtype=torch.float32) for m in ms]))
        t = _abi.MotionTables()
        t.num_motions = len(ms)
        for k, v in self._tabs.items():
            setattr(t, k, v.data_ptr())
        self._tables = t
