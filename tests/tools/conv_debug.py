# Debug aid (imports the oracle as the checker, hence it lives under tests/, not tools/).
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from oracle.conv_rows import conv1d_rows
from speakerguard_amd import _native as N
import test_gpu_conv as T
ctx = N.Context()
B, Ta, Tc, Kc, n, taps, step, base = 64, 270, 266, 192, 512, 3, 2, 0
a, w = T._case(7, B, Ta, Tc, Kc, n, taps)
want = conv1d_rows(a, w, B, Ta, Tc, taps, step, base)
for k in (0, 2):
    o = T._run(ctx, a, w, B, Ta, Tc, taps, step, base, 0, k)
    bad = np.abs(o - want) > 1e-3
    print("kernel", k, "bad frac", bad.mean())
    if bad.any():
        tiles = bad.reshape(133, 128, 4, 128).any(axis=(1, 3))
        print("bad tiles (mt, nt):", np.argwhere(tiles)[:40].tolist(), "count", tiles.sum())
        mt, nt = np.argwhere(tiles)[0]
        sub = bad[mt * 128:(mt + 1) * 128, nt * 128:(nt + 1) * 128]
        print("in first bad tile: bad rows", np.nonzero(sub.any(1))[0].tolist()[:140])
        print("bad cols", np.nonzero(sub.any(0))[0].tolist()[:140])
