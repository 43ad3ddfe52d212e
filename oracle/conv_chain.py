"""TEST INFRASTRUCTURE ONLY -- ctypes loader of oracle/conv_chain.c (the contraction kernels' arithmetic as one float32
fmaf chain per output, bit for bit; see the C file).  Built by ``make oracle`` / ``__graft_entry__.build()``."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libconv_chain.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "conv_chain.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(["gcc", "-O2", "-mfma", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", _SO, "-lm"], check=True)
        _lib = C.CDLL(_SO)
        _lib.sg_conv_chain.restype = None
        _lib.sg_feco_scores.restype = None
    return _lib


def conv_chain(a, w, B, Ta, Tc, taps, tap_step, tap_base=0, bias=None, mask=None):
    """a (B*Ta, Kc), w (taps*Kc, N) float32 -> (B*Tc, N) float32: the kernels' fmaf chain (Kc a multiple of 32)."""
    a = np.ascontiguousarray(a, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    Kc, N = a.shape[1], w.shape[1]
    assert Kc % 32 == 0 and w.shape[0] == taps * Kc and a.shape[0] == B * Ta
    out = np.empty((B * Tc, N), np.float32)
    fp = lambda x: None if x is None else np.ascontiguousarray(x, np.float32).ctypes.data_as(C.c_void_p)
    keep = [np.ascontiguousarray(x, np.float32) for x in (bias, mask) if x is not None]  # noqa: F841 (lifetime)
    epi = 1 if bias is not None else (2 if mask is not None else 0)
    _load().sg_conv_chain(a.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), fp(bias), fp(mask),
                          out.ctypes.data_as(C.c_void_p), B, Ta, Tc, Kc, N, taps, tap_step, tap_base, epi)
    return out


def feco_scores(xc, cc, h):
    """FeCo assignment scores (k_feco.hip contract version 2): xc (F, Dp), cc (k, Dp), h (k) float32 -> (F, k) float32,
    score(i, j) = fmaf chain from h[j] over the kernels' k order (Dp = 32 or 64, pad dimensions zero)."""
    xc = np.ascontiguousarray(xc, np.float32)
    cc = np.ascontiguousarray(cc, np.float32)
    h = np.ascontiguousarray(h, np.float32)
    F, Dp = xc.shape
    k = cc.shape[0]
    assert Dp in (32, 64) and cc.shape[1] == Dp and h.shape == (k,)
    out = np.empty((F, k), np.float32)
    _load().sg_feco_scores(xc.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p),
                           out.ctypes.data_as(C.c_void_p), F, k, Dp)
    return out
