"""A/B of the FeCo k-means on one vs two compute units per instance (sg_feco_set_two_cu): the configs[3] clustering call
(64 utterances x 2 repeats, 300 x 32 log-mel frames of real AudioNet features, 150 clusters, random start), 200 calls per mode,
modes alternating."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import _native as N, synth
from speakerguard_amd.model.audionet_csine import audionet_csine
dev = torch.device("cuda:0")
m = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=dev)
B, reps = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 2
x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=3)).to(dev)
feat = m.compute_feat(x).contiguous()
F, D = feat.shape[1], feat.shape[2]
k = F // 2
ids = torch.empty(reps * B, F, device=dev, dtype=torch.int32)
out = torch.empty(reps * B, k, D, device=dev)
counts = torch.empty(reps * B, k, device=dev, dtype=torch.int32)
ctx = m.ctx
def run(n, seed0):
    for i in range(n):
        ctx.call("sg_feco_kmeans_compress", N._ptr(feat), B, F, D, k, 10, 1, seed0 + i, 0, reps, N._ptr(ids), N._ptr(out), N._ptr(counts),
                 N.current_stream_ptr(dev))
res = {0: [], -1: []}
for rnd in range(4):
    for mode in (0, -1):
        ctx.call("sg_feco_set_two_cu", mode)
        run(5, 1000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(50, 7)
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / 50 * 1e6)
for mode, name in ((0, "one compute unit per instance"), (-1, "two where they fit")):
    print("B=%d x %d repeats, %d x %d frames, k=%d: %s: %s us per call" % (B, reps, F, D, k, name, " ".join("%.1f" % v for v in res[mode])))
