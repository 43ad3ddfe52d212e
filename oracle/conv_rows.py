"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- numpy restatement of the dilated Conv1d contraction.

Forward: torch.nn.functional.conv1d as used by the TDNN (reference model/_xv_plda/xvecTDNN.py:16-33: five
Conv1d with kernel 5/3/3/1/1, dilation 1/2/3/1/1, no padding), written on channel-last rows; the data
gradient autograd derives for it (reference adaptive_attack/EOT.py:35, loss.backward) is the same contraction
over d(out) with a negative tap step and rows outside the utterance contributing zero.  Accumulates in
float64 so it can serve as the "true" value when two float32 summation orders are compared.
"""
import numpy as np


def conv1d_rows(a, w, B, Ta, Tc, taps, tap_step, tap_base=0, bias=None, mask=None):
    """a: (B*Ta, Kc), w: (taps*Kc, N) -> (B*Tc, N) float64.

    out[b*Tc + t] = sum_j a[b*Ta + t + tap_base + j*tap_step] @ w[j*Kc:(j+1)*Kc]; rows outside [0, Ta)
    contribute zero; then relu(out + bias) if bias is given, or out * (mask > 0) if mask is given.
    """
    Kc = a.shape[1]
    N = w.shape[1]
    a3 = a.reshape(B, Ta, Kc).astype(np.float64)
    w64 = w.astype(np.float64)
    out = np.zeros((B, Tc, N))
    t = np.arange(Tc)
    for j in range(taps):
        src = t + tap_base + j * tap_step
        ok = (src >= 0) & (src < Ta)
        if not ok.any():
            continue
        out[:, ok] += a3[:, src[ok]] @ w64[j * Kc:(j + 1) * Kc]
    out = out.reshape(B * Tc, N)
    if bias is not None:
        out = np.maximum(out + bias.astype(np.float64), 0.0)
    if mask is not None:
        out = np.where(mask > 0, out, 0.0)
    return out


def conv1d_torch_layout(x, weight, dilation):
    """The same forward contraction from torch's layout: x (B, Cin, T), weight (Cout, Cin, k) -> (B, Cout, T')."""
    B, Cin, T = x.shape
    Cout, _, k = weight.shape
    Tc = T - (k - 1) * dilation
    a = np.ascontiguousarray(x.transpose(0, 2, 1)).reshape(B * T, Cin)
    w = np.ascontiguousarray(weight.transpose(2, 1, 0)).reshape(k * Cin, Cout)
    return conv1d_rows(a, w, B, T, Tc, k, dilation).reshape(B, Tc, Cout).transpose(0, 2, 1)
