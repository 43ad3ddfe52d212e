"""A/B of the AudioNet front-end settings (sg_an_configure): transform precision x spectrum cache x overlap-add inside the
adjoint, PGD-20 at 64 and 512 utterances of 3 s.  argv: optional "bits,cache,ola" triples and batch sizes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.audionet_csine import audionet_csine
dev = torch.device("cuda:0")
m = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=dev)
args = [a for a in sys.argv[1:] if "," in a]
batches = [int(a) for a in sys.argv[1:] if "," not in a] or [64, 512]
cfgs = [tuple(int(v) for v in a.split(",")) for a in args] or [(64, 0, 0), (64, 1, 0), (64, 0, 1), (32, 0, 0), (32, 1, 0), (32, 0, 1), (32, 1, 1)]
spec = SEC4SR_CrossEntropy()
for B in batches:
    x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=5)).to(dev)
    y = m.make_decision(x)[0]
    lo, hi = torch.clamp(x - 0.002, min=-1), torch.clamp(x + 0.002, max=1)
    for bits, cache, ola in cfgs:
        m.configure_frontend(bits, None if cache < 0 else bool(cache), None if ola < 0 else bool(ola))  # -1: the engine's own choice
        m.pgd_run(x, y, lo, hi, spec, 0.0004, 3, 1)
        torch.cuda.synchronize()
        best = 1e9
        K = 20
        for _ in range(3):
            t0 = time.perf_counter()
            m.pgd_run(x, y, lo, hi, spec, 0.0004, K, 1)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("B=%3d  fft %d  spectrum cache %d  fused overlap-add %d: %.3f ms per PGD step" % (B, bits, cache, ola, 1e3 * best / K), flush=True)
