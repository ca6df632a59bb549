"""ctypes front end of the CPU checker (oracle/libigx_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
Arrays are C-contiguous numpy arrays, modified in place like the device
buffers of the library under test.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from isaacgym_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libigx_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "igx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_philox_uniform.restype = C.c_float
        _lib.orc_philox_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int]
    return _lib


def _np_ptr(a):
    if not isinstance(a, np.ndarray) or not a.flags["C_CONTIGUOUS"]:
        raise TypeError("oracle buffers must be C-contiguous numpy arrays")
    return a.ctypes.data


def _call(fn, params, bufcls, bufs, n):
    b = _abi.make_buffers(bufcls, bufs, _np_ptr)
    rc = fn(C.byref(params), C.byref(b), C.c_int64(n))
    if rc != 0:
        raise RuntimeError("oracle call failed: %d" % rc)


def set_num_threads(n):
    """Set (n > 0) and return the number of host threads the checker uses."""
    return int(lib().orc_set_num_threads(int(n)))


def philox_uniform(seed, offset, env, j):
    return float(lib().orc_philox_uniform(seed, offset, env, j))


def cartpole_step(params, bufs, n):
    _call(lib().orc_cartpole_step, params, _abi.CartpoleBuffers, bufs, n)


def ant_step(params, bufs, n):
    _call(lib().orc_ant_step, params, _abi.LocoBuffers, bufs, n)


def humanoid_step(params, bufs, n):
    _call(lib().orc_humanoid_step, params, _abi.LocoBuffers, bufs, n)
