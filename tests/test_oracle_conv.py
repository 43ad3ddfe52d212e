"""CPU check of the conv oracle (oracle/conv_rows.py) against torch's own conv1d and its autograd data gradient."""
import numpy as np
import torch

from oracle.conv_rows import conv1d_rows, conv1d_torch_layout


def test_forward_equals_torch_conv1d():
    rng = np.random.RandomState(0)
    for cin, cout, k, dil, T in ((30, 64, 5, 1, 40), (64, 32, 3, 2, 31), (32, 32, 3, 3, 50), (16, 8, 1, 1, 9)):
        x = rng.standard_normal((3, cin, T))
        w = rng.standard_normal((cout, cin, k))
        ref = torch.nn.functional.conv1d(torch.from_numpy(x), torch.from_numpy(w), dilation=dil).numpy()
        assert np.abs(conv1d_torch_layout(x, w, dil) - ref).max() < 1e-11


def test_data_gradient_equals_autograd():
    """d/dx of sum(conv1d(x, w) * g) is the same contraction over g with tap step -dilation."""
    rng = np.random.RandomState(1)
    B, cin, cout, k, dil, T = 2, 32, 64, 3, 2, 29
    Tc = T - (k - 1) * dil
    x = torch.from_numpy(rng.standard_normal((B, cin, T))).requires_grad_(True)
    w = torch.from_numpy(rng.standard_normal((cout, cin, k)))
    g = rng.standard_normal((B, cout, Tc))
    (torch.nn.functional.conv1d(x, w, dilation=dil) * torch.from_numpy(g)).sum().backward()
    a = np.ascontiguousarray(g.transpose(0, 2, 1)).reshape(B * Tc, cout)           # d(out) rows
    wb = np.ascontiguousarray(w.numpy().transpose(2, 0, 1)).reshape(k * cout, cin)  # [j][co][ci]
    # dx[t] = sum_j g[t - j*dil] @ w[:, :, j]  -> tap_step = -dil, rows outside [0, Tc) are zero
    dx = conv1d_rows(a, wb, B, Tc, T, k, -dil).reshape(B, T, cin).transpose(0, 2, 1)
    assert np.abs(dx - x.grad.numpy()).max() < 1e-11
