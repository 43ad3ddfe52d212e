"""Reads the per-worker phase timestamps the stream-K kernel writes when SG_SK_TRACE=<file> is set
(100 MHz wall clock, 16 slots per worker, see conv_gemm_streamk_kernel) and prints where the time goes.

    SG_SK_TRACE=gpurun_out/sk.bin python tools/layer_bench.py --layers 5 --iters 2 --repeats 1
    python tools/sk_trace.py gpurun_out/sk.bin
    python tools/sk_trace.py gpurun_out/sk.bin --all     one line per traced launch: shape, busy time, shader clock, chunk time
                                                         (with SG_SK_TRACE_SKIP / SG_SK_TRACE_COUNT: launches inside a long loop)
"""
import struct
import sys
from collections import Counter

import numpy as np

data = open(sys.argv[1], "rb").read()
pos, recs = 0, []
while pos < len(data):
    hdr = struct.unpack("8i", data[pos:pos + 32]); pos += 32
    n = hdr[0] * 16
    t = np.frombuffer(data[pos:pos + 8 * n], dtype=np.uint64).reshape(hdr[0], 16).copy(); pos += 8 * n
    recs.append((hdr, t))
if "--all" in sys.argv:
    for hdr, t in recs:
        workers, M, N, C, ipw, tiles, epi, _ = hdr
        xcc = ((t[:, 15] >> np.uint64(32)).astype(np.int64)) & 0xF
        ts = t[:, :15].astype(np.float64) * 0.01
        def colr(i):
            v = ts[:, i].copy(); v[t[:, i] == 0] = np.nan; return v
        start = colr(0)
        end = np.nanmax(np.stack([colr(i) for i in (2, 4, 6, 8, 12)]), axis=0)
        busy = end - start
        cyc = t[:, 14].astype(np.float64) - t[:, 13].astype(np.float64)
        okc = (t[:, 13] != 0) & (t[:, 14] > t[:, 13]) & ~np.isnan(busy) & (busy > 0)
        mhz = np.median(cyc[okc] / busy[okc]) if okc.sum() > 8 else float("nan")
        whole = colr(5) - colr(4)  # second whole tile: slots 4 (end of tile 0's epilogue) -> 5 (end of tile 1's chunks)
        whole = whole[~np.isnan(whole) & (whole > 0)]
        if not whole.size:          # long tiles: the first whole tile, from the end of the parked head piece (or the start)
            t0 = np.where(np.isnan(colr(2)), colr(0), colr(2))
            whole = colr(3) - t0
            whole = whole[~np.isnan(whole) & (whole > 0)]
        per_chunk = np.median(whole) / C if whole.size else float("nan")
        tile_rows = M * 1.0 / (tiles / (N // 128))
        flop_chunk = 2.0 * round(tile_rows / 32) * 32 * 128 * 32
        ideal = flop_chunk / (256 * mhz) if mhz == mhz else float("nan")  # 157.3 TFLOP/s = 256 CUs x 256 flop per clock at 2.4 GHz
        print("M %6d N %5d chunks/tile %3d epi %d workers %d: busy median %7.1f us  clock %4.0f MHz  whole tile %.3f us per chunk (ideal at that clock %.3f: %.1f %%)" % (
            M, N, C, epi, workers, np.nanmedian(busy), mhz, per_chunk, ideal, 100 * ideal / per_chunk if per_chunk == per_chunk else float("nan")))
    sys.exit(0)
hdr, t = recs[-1]
workers, M, N, C, ipw, tiles, epi, _ = hdr
print("launch: workers %d  M %d N %d  chunks/tile %d  chunks/worker %d  tiles %d  epi %d" % (workers, M, N, C, ipw, tiles, epi))
hw = t[:, 15]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(np.int64)
sh = ((hw >> np.uint64(12)) & np.uint64(0x1)).astype(np.int64)
se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
place = xcc * 1000 + se * 100 + sh * 10 + cu
cnt = Counter(place.tolist())
print("distinct (xcc,se,sh,cu) places: %d; blocks per place histogram: %s" % (len(cnt), Counter(cnt.values())))
by_place = {}
for w in range(workers):
    by_place.setdefault(int(place[w]), []).append(w)
pairs = [v for v in by_place.values() if len(v) == 2]
d = [abs(a - b) for a, b in pairs]
print("worker-index distance of co-resident pairs: %s" % Counter(d).most_common(6))
print("xcc of workers 0..15:", xcc[:16].tolist(), " workers 60..70:", xcc[60:70].tolist())
ts = t[:, :15].astype(np.float64) * 0.01  # us
# the 100 MHz counters of different XCDs are not aligned: normalise per XCD
for x in range(8):
    sel = xcc == x
    if sel.any():
        ts[sel] -= ts[sel, 0].min()
def col(i):
    v = ts[:, i].copy(); v[t[:, i] == 0] = np.nan; return v
start = col(0)
print("start skew inside an XCD: median %.1f p90 %.1f max %.1f us" % (np.nanmedian(start), np.nanpercentile(start, 90), np.nanmax(start)))
end = np.nanmax(np.stack([col(i) for i in (2, 4, 6, 8, 12)]), axis=0)
busy = end - start
print("per-worker start->end: median %.1f p10 %.1f p90 %.1f max %.1f us;  last end inside its XCD: %.1f us" % (
    np.nanmedian(busy), np.nanpercentile(busy, 10), np.nanpercentile(busy, 90), np.nanmax(busy), np.nanmax(end)))
cyc = t[:, 14].astype(np.float64) - t[:, 13].astype(np.float64)
okc = (t[:, 13] != 0) & (t[:, 14] > t[:, 13]) & ~np.isnan(busy) & (busy > 0)
if okc.sum() > 8:
    mhz = cyc[okc] / busy[okc]
    print("shader clock over the worker's busy time (s_memtime ticks per us): median %.0f MHz  p10 %.0f  p90 %.0f;  by XCD: %s" % (
        np.median(mhz), np.percentile(mhz, 10), np.percentile(mhz, 90),
        ["%d: %.0f" % (x, np.median(mhz[xcc[okc] == x])) for x in range(8) if (xcc[okc] == x).any()]))
chunk_us = 128 * 128 * 32 * 2 * 2 / (157.3e12 / 256) * 1e6
print("ideal chunk time at peak with 2 blocks per CU: %.2f us" % chunk_us)
def rep(name, a, b, chunks=None):
    dlt = col(b) - col(a)
    ok = ~np.isnan(dlt)
    if ok.sum() == 0:
        return
    msg = "%-28s n=%3d  median %7.2f us  p90 %7.2f" % (name, ok.sum(), np.nanmedian(dlt), np.nanpercentile(dlt[ok], 90))
    if chunks is not None:
        per = dlt[ok] / chunks[ok]
        msg += "   per chunk %.2f us" % np.nanmedian(per)
    print(msg)
w = np.arange(workers)
it_begin = w * ipw
it_end = np.minimum(tiles * C, it_begin + ipw)
first_c0 = it_begin % C
last_c1 = (it_end - 1) % C + 1
rep("head piece compute", 0, 1, last_c1.astype(float))
rep("park slab + flag", 1, 2)
rep("whole tile 0 compute", 2, 3, np.full(workers, C, float))
rep("whole tile 0 epilogue", 3, 4)
rep("whole tile 1 compute", 4, 5, np.full(workers, C, float))
rep("whole tile 1 epilogue", 5, 6)
rep("flag wait + acquire", 8, 9)
rep("slab read", 9, 10)
rep("tail piece compute", 10, 11, (C - first_c0).astype(float))
rep("  tail: set-up + first loads -> prologue done", 10, 13)
rep("  tail: first two chunks", 13, 14)
rep("  tail: remaining chunks", 14, 11, (C - first_c0 - 2).astype(float))
rep("tail epilogue", 11, 12)
print()
print("busy time by XCD (median / max us):", ["%d: %.0f/%.0f" % (x, np.nanmedian(busy[xcc == x]), np.nanmax(busy[xcc == x])) for x in range(8) if (xcc == x).any()])
occ = np.array([cnt[int(pl)] for pl in place])
for k in sorted(set(occ.tolist())):
    print("blocks on CUs hosting %d blocks: n=%d  busy median %.0f  max %.0f us" % (k, (occ == k).sum(), np.nanmedian(busy[occ == k]), np.nanmax(busy[occ == k])))
order = np.argsort(busy)
print("slowest 12 workers:", [(int(i), int(xcc[i]), int(place[i]) % 1000, int(occ[i]), round(float(busy[i]))) for i in order[-12:]])
print("fastest 6 workers:", [(int(i), int(xcc[i]), int(place[i]) % 1000, int(occ[i]), round(float(busy[i]))) for i in order[:6]])
hist, edges = np.histogram(busy[~np.isnan(busy)], bins=12)
print("busy histogram:", list(zip([round(float(e)) for e in edges[:-1]], hist.tolist())))
pa = np.array([[busy[a], busy[b]] for a, b in pairs if not (np.isnan(busy[a]) or np.isnan(busy[b]))])
if pa.ndim != 2 or not len(pa):
    print("no CU hosts two blocks: nothing to compare")
    sys.exit(0)
lo, hi = pa.min(1), pa.max(1)
print("co-resident pairs: faster block median %.0f us, slower block median %.0f us; pairs with both < 430: %d, both > 430: %d, mixed: %d" % (
    np.median(lo), np.median(hi), (hi < 430).sum(), (lo > 430).sum(), ((lo < 430) & (hi > 430)).sum()))
first = np.array([min(a, b) for a, b in pairs])  # lower worker index of the pair
fast_is_lower = np.mean([busy[min(a, b)] < busy[max(a, b)] for a, b in pairs])
print("fraction of pairs where the lower worker index (first dispatched) is the faster one: %.2f" % fast_is_lower)
