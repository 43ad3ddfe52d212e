"""CPU check of the conv oracle (oracle/conv_rows.py) against torch's own conv1d and its autograd data gradient."""
import numpy as np
import torch

from oracle.conv_chain import conv_chain
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


def test_fmaf_chain_restatement_of_the_kernel_arithmetic():
    """oracle/conv_chain.c (the contraction kernels' arithmetic: one float32 fmaf chain per output in the kernels' k order,
    which the GPU tests compare BIT FOR BIT with every launch strategy) agrees with the float64 contraction to float32
    round-off, applies the epilogues, treats rows outside the utterance as zeros, and is a pure function of its inputs."""
    rng = np.random.RandomState(2)
    for B, Ta, Tc, Kc, N, taps, step, base in ((3, 50, 46, 64, 128, 5, 1, 0), (2, 40, 46, 96, 256, 3, -3, 0), (5, 33, 33, 32, 128, 3, 1, -1)):
        a = rng.standard_normal((B * Ta, Kc)).astype(np.float32)
        w = (rng.standard_normal((taps * Kc, N)) / np.sqrt(taps * Kc)).astype(np.float32)
        want = conv1d_rows(a, w, B, Ta, Tc, taps, step, base)
        got = conv_chain(a, w, B, Ta, Tc, taps, step, base)
        assert got.dtype == np.float32 and np.abs(got - want).max() / np.abs(want).max() < 2e-6
        assert np.array_equal(got.view(np.uint32), conv_chain(a, w, B, Ta, Tc, taps, step, base).view(np.uint32))
        bias = rng.standard_normal(N).astype(np.float32)
        mask = (rng.standard_normal((B * Tc, N)) > 0).astype(np.float32)
        assert np.array_equal(conv_chain(a, w, B, Ta, Tc, taps, step, base, bias=bias), np.maximum(got + bias, np.float32(0)))
        assert np.array_equal(conv_chain(a, w, B, Ta, Tc, taps, step, base, mask=mask), np.where(mask > 0, got, np.float32(0)))
    # the k order matters at the last bit: the natural order 0..7 gives other bits somewhere
    a = rng.standard_normal((64, 64)).astype(np.float32)
    w = rng.standard_normal((64, 128)).astype(np.float32)
    nat = np.zeros((64, 128), np.float32)
    for k in range(64):
        nat = (nat.astype(np.float64) + a[:, k:k + 1].astype(np.float64) * w[k:k + 1].astype(np.float64)).astype(np.float32)  # exact product, one rounding = fmaf
    assert not np.array_equal(nat, conv_chain(a, w, 1, 64, 64, 1, 1, 0))
