"""Equal-error-rate threshold; mirrors reference set_threshold.py:22-47 ``set_threshold``.

The O(n_target * (n_target + n_untarget)) scan runs on the device (C-ABI ``sg_eer_threshold``) and returns
the same element of ``score_target`` the reference loop picks (first minimum of |FRR - FAR|).
"""
import numpy as np
import torch

from . import _native as N
from .metric.metric import _context


def set_threshold(score_target, score_untarget, device="cuda:0"):
    """-> (final_threshold, final_frr, final_far) with FRR / FAR in percent."""
    dev = torch.device(device)
    st = torch.as_tensor(np.asarray(score_target, dtype=np.float32)).flatten().to(dev).contiguous()
    su = torch.as_tensor(np.asarray(score_untarget, dtype=np.float32)).flatten().to(dev).contiguous()
    out = torch.empty(3, device=dev, dtype=torch.float64)
    _context(dev).call("sg_eer_threshold", N._ptr(st), st.numel(), N._ptr(su), su.numel(), N._ptr(out),
                       N.current_stream_ptr(dev))
    thr, frr, far = out.cpu().tolist()
    return thr, frr, far
