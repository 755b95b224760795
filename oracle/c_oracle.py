"""TEST INFRASTRUCTURE ONLY - ctypes access to oracle/_build/liboracle.so (gae_oracle.c).

The plain-C restatement exists to check the HIP kernels bit for bit (one IEEE fp32 rounding
per operation, no FMA contraction) and as an implementation independent of PyTorch.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle.so')
_lib = None


def build():
    subprocess.run(['make', '-C', _HERE], check=True, capture_output=True)


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def gae_f32_scan(rewards, values, dones, last_values, last_dones, gamma, tau):
    """numpy fp32 arrays, time-major [H,N,V] / [H,N] / [N,V] / [N].  Returns advs [H,N,V]."""
    lib = _load()
    r = np.ascontiguousarray(rewards, dtype=np.float32)
    v = np.ascontiguousarray(values, dtype=np.float32)
    d = np.ascontiguousarray(dones, dtype=np.float32)
    lv = np.ascontiguousarray(last_values, dtype=np.float32)
    ld = np.ascontiguousarray(last_dones, dtype=np.float32)
    H, N, V = r.shape
    out = np.empty_like(r)
    lib.gae_f32_scan(_fp(r), _fp(v), _fp(d), _fp(lv), _fp(ld), _fp(out),
                     ctypes.c_int(H), ctypes.c_int(N), ctypes.c_int(V),
                     ctypes.c_float(np.float32(gamma)),
                     ctypes.c_float(np.float32(float(gamma) * float(tau))))
    return out


def returns_and_advantages(advs, values):
    lib = _load()
    a = np.ascontiguousarray(advs, dtype=np.float32)
    v = np.ascontiguousarray(values, dtype=np.float32)
    ret = np.empty_like(a)
    adv = np.empty_like(a)
    lib.returns_and_advantages_f32(_fp(a), _fp(v), _fp(ret), _fp(adv), ctypes.c_size_t(a.size))
    return ret, adv


def gae_f64_reference(rewards, values, dones, last_values, last_dones, gamma, tau):
    lib = _load()
    r = np.ascontiguousarray(rewards, dtype=np.float32)
    v = np.ascontiguousarray(values, dtype=np.float32)
    d = np.ascontiguousarray(dones, dtype=np.float32)
    lv = np.ascontiguousarray(last_values, dtype=np.float32)
    ld = np.ascontiguousarray(last_dones, dtype=np.float32)
    H, N, V = r.shape
    out = np.empty(r.shape, dtype=np.float64)
    lib.gae_f64_reference(_fp(r), _fp(v), _fp(d), _fp(lv), _fp(ld),
                          out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                          ctypes.c_int(H), ctypes.c_int(N), ctypes.c_int(V),
                          ctypes.c_double(gamma), ctypes.c_double(tau))
    return out
