"""TrajectoryRecorder: stream selected observation columns of every environment to
pinned host memory, one capture per step, on a side stream (libigx igx_recorder_*).

Counterpart of the sampling mode of the reference's tasks/cartpole_sample.py
(:99-107 start, :119-121 get_x_theta, :289-293 per-step torch.cat, :286 torch.save):
the file it writes is the same `torch.save` of a dict name -> float32 tensor
[num_envs, steps] (for CartpoleSample: "x" and "theta", T + 1 columns).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


class TrajectoryRecorder:
    def __init__(self, num_envs, row_width, columns, max_steps, device, ring_slots=4):
        """columns: dict name -> column index of the recorded [num_envs, row_width] buffer, or None to record
        whole rows (finish() then returns one [steps, num_envs, row_width] tensor: the batches of
        HumanoidAMP.fetch_amp_obs_demo, streamed to the host as they are sampled)."""
        self.names = None if columns is None else list(columns)
        self.num_envs, self.row_width, self.max_steps = int(num_envs), int(row_width), int(max_steps)
        self.device = torch.device(device)
        if self.names is None:
            cols, ncols = None, 0
        else:
            cols, ncols = (C.c_int32 * len(self.names))(*[int(columns[k]) for k in self.names]), len(self.names)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().igx_recorder_create(C.byref(self._h), self.num_envs, self.row_width, cols,
                                                      ncols, int(ring_slots), self.max_steps))

    def capture(self, buf):
        """Record the current contents of `buf` ([num_envs, row_width] fp32 on the device); asynchronous."""
        if tuple(buf.shape) != (self.num_envs, self.row_width) or not buf.is_contiguous():
            raise ValueError("expected a [%d, %d] buffer, got %s" % (self.num_envs, self.row_width, tuple(buf.shape)))
        if buf.device != self.device:
            raise ValueError("buffer is on %s, recorder on %s" % (buf.device, self.device))
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().igx_recorder_capture(self._h, _lib.tensor_ptr(buf, torch.float32),
                                                       _lib.current_stream(self.device)))

    @property
    def steps(self):
        return int(_lib.lib().igx_recorder_steps(self._h))

    def finish(self):
        """Wait for outstanding copies; returns dict name -> torch tensor [num_envs, steps] (a copy)."""
        _lib.check(_lib.lib().igx_recorder_sync(self._h))
        steps = self.steps
        ptr = _lib.lib().igx_recorder_host_data(self._h)
        if self.names is None:  # whole rows: [steps, num_envs, row_width]
            n = steps * self.num_envs * self.row_width
            if n == 0:
                return torch.zeros(0, self.num_envs, self.row_width)
            flat = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n,))
            return torch.from_numpy(flat.reshape(steps, self.num_envs, self.row_width).copy())
        n = steps * len(self.names) * self.num_envs
        if n == 0:
            return {k: torch.zeros(self.num_envs, 0) for k in self.names}
        flat = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n,))
        arr = flat.reshape(steps, len(self.names), self.num_envs)
        return {k: torch.from_numpy(np.ascontiguousarray(arr[:, i, :].T)) for i, k in enumerate(self.names)}

    def save(self, path):
        out = self.finish()
        torch.save(dict(out) if self.names is not None else {"rows": out}, path)

    def close(self):
        if self._h:
            _lib.lib().igx_recorder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_sequence(path):
    """Read a sequence file (this recorder's or the reference's own): tensors only, nothing executed."""
    return torch.load(path, weights_only=True)
