"""Perturbation metrics of the evaluation step; mirrors reference metric/metric.py:8-75.

L2 / L0 / L1 / Linf / SNR of a batch come from ONE native pass (C-ABI ``sg_wav_finalize``), with the
reference's ``preprocess`` rule (divide by 2^15 unless -1 <= max <= 1, metric.py:8-12) applied per utterance.
PESQ and STOI are third-party packages (pesq, pystoi) outside the accelerated path: they raise.
"""
import ctypes as C

import numpy as np
import torch

from .. import _native as N

_ctx = {}


def _context(device):
    """One engine context per GPU.  An index-less ``cuda`` device means torch's CURRENT device (not GPU 0)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise N.NativeError("the HIP engine needs a GPU tensor/device (got %s)" % device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ctx:
        _ctx[idx] = N.Context(idx)
    return _ctx[idx]


def _rows(x):
    x = x.detach() if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    x = x.to(torch.float32)
    return x.reshape(1, -1) if x.dim() <= 1 or (x.dim() == 2 and x.shape[0] == 1) else x.reshape(x.shape[0], -1)


def batch_metrics(benign, adver, device=None):
    """(N,1,T) or (1,T) pairs -> float64 array (N,5): L2, L0, L1, Linf, SNR (metric.py:70-75 order)."""
    b, a = _rows(benign), _rows(adver)
    if b.shape != a.shape:
        raise ValueError("benign and adversarial audio must have the same shape, got %s vs %s" % (tuple(b.shape), tuple(a.shape)))
    dev = torch.device(device) if device is not None else (a.device if a.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    if dev.type != "cuda":
        raise N.NativeError("metrics run on the HIP device only (got %s)" % dev)
    b, a = b.to(dev).contiguous(), a.to(dev).contiguous()
    out = torch.empty(a.shape[0], 5, device=dev, dtype=torch.float64)
    _context(dev).call("sg_wav_finalize", N._ptr(b), N._ptr(a), a.shape[0], a.shape[1], None, N._ptr(out),
                       N.current_stream_ptr(dev))
    return out.cpu().numpy()


def _one(benign_xx, adver_xx, k):
    return float(batch_metrics(benign_xx, adver_xx)[0, k])


def L2(benign_xx, adver_xx, bits=16):
    return _one(benign_xx, adver_xx, 0)


def L0(benign_xx, adver_xx, bits=16):
    return _one(benign_xx, adver_xx, 1)


def L1(benign_xx, adver_xx, bits=16):
    return _one(benign_xx, adver_xx, 2)


def Linf(benign_xx, adver_xx, bits=16):
    return _one(benign_xx, adver_xx, 3)


def SNR(benign_xx, adver_xx, bits=16):
    return _one(benign_xx, adver_xx, 4)


def PESQ(benign_xx, adver_xx, bits=16):
    raise NotImplementedError("PESQ needs the third-party 'pesq' package (metric.py:44-48); not part of the native path")


def STOI(benign_xx, adver_xx, fs=16_000, bits=16):
    raise NotImplementedError("STOI needs the third-party 'pystoi' package (metric.py:50-54); not part of the native path")


def get_all_metric(benign_xx, adver_xx, fs=16_000, bits=16):
    """[L2, L0, L1, Linf, SNR] of one utterance (metric.py:56-64 without the PESQ / STOI entries)."""
    return [float(v) for v in batch_metrics(benign_xx, adver_xx)[0]]
