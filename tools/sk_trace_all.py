"""Per-launch summary of a stream-K phase trace taken over a whole attack (SG_SK_TRACE=<file> python tools/step_profile.py 64 3):
segment / epilogue / hand-off medians of every launch, in launch order, to compare in-loop launches with isolated ones."""
import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read()
pos, n = 0, 0
while pos < len(data):
    hdr = struct.unpack("8i", data[pos:pos + 32]); pos += 32
    workers, M, N, C, ipw, tiles, epi, _ = hdr
    t = np.frombuffer(data[pos:pos + 8 * workers * 16], dtype=np.uint64).reshape(workers, 16).astype(np.float64) * 0.01; pos += 8 * workers * 16
    def d(a, b):
        v = t[:, b] - t[:, a]; ok = (t[:, a] > 0) & (t[:, b] > 0)
        return np.median(v[ok]) if ok.any() else float("nan")
    hw = np.frombuffer(data[pos - 8 * workers * 16:pos], dtype=np.uint64).reshape(workers, 16)[:, 15]
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    span = []
    for x in range(8):
        sel = xcc == x
        if sel.any():
            ends = np.max(t[sel][:, [2, 4, 6, 8, 12]], axis=1)
            span.append((ends - t[sel, 0].min()).max())
    print("launch %3d epi %d M %6d N %5d chunks %3d workers %3d: span %7.1f us | park %.2f whole-epi %.2f | wait+slab %.2f tail-epi %.2f" % (
        n, epi, M, N, C, workers, max(span), d(1, 2), d(3, 4), d(9, 10), d(11, 12)))
    n += 1
