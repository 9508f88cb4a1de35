"""ctypes wrapper of the C oracle (oracle/_build/libsid_oracle.so).  TEST INFRASTRUCTURE ONLY:
checker for tests/ and the timed CPU baseline of bench.py; never part of the product path."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libsid_oracle.so")
_fp = ctypes.POINTER(ctypes.c_float)
_lib = None


def build():
    subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
        _lib.sid_oracle_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_fp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def threads():
    return load().sid_oracle_threads()


def injected_attention_rows(row0, nrows, xq, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt, m, gamma, T, scale, q_self_coef=0.5):
    H, N, d = q_cnt.shape
    arrs = [_f32(a) for a in (xq, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt, m)]
    out = np.empty((nrows, H * d), dtype=np.float32)
    f = ctypes.c_float
    load().sid_oracle_injected_attn_rows(*[_p(a) for a in arrs], _p(out), H, N, d, f(gamma), f(q_self_coef), f(T),
                                         f(scale), int(row0), int(nrows))
    return out


def plain_attention_rows(row0, nrows, q, k, v, scale, T=1.0):
    H, Nq, d = q.shape
    Nk = k.shape[1]
    q, k, v = _f32(q), _f32(k), _f32(v)
    out = np.empty((nrows, H * d), dtype=np.float32)
    f = ctypes.c_float
    load().sid_oracle_plain_attn_rows(_p(q), _p(k), _p(v), _p(out), H, Nq, Nk, d, f(T), f(scale), int(row0), int(nrows))
    return out


def mask_downsample(mask, out_h, out_w, ch=1.0):
    mask = _f32(mask)
    out = np.empty(out_h * out_w, dtype=np.float32)
    load().sid_oracle_mask_downsample(_p(mask), mask.shape[0], mask.shape[1], out_h, out_w, ctypes.c_float(ch), _p(out))
    return out


def hf_filter(h_cnt, ratio=0.5, skip=None, h=None):
    h_cnt = _f32(h_cnt)
    B, C, H, W = h_cnt.shape
    skip = None if skip is None else _f32(skip)
    h = None if h is None else _f32(h)
    out = np.empty_like(h_cnt)
    load().sid_oracle_hf_filter(_p(h_cnt), _p(skip), _p(h), _p(out), B * C, H, W, int(min(H, W) * ratio))
    return out
