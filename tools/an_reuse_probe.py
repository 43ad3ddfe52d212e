"""sg_an_logmel_backward with reuse_forward = 1 against 0: how far apart are the two adjoints, and where?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import _native as N, synth
from speakerguard_amd.model.audionet_csine import audionet_csine
dev = torch.device("cuda:0")
hip = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=dev)
for bits in (32, 64):
    for cache in (True, False):
        hip.configure_frontend(bits, cache, False)
        x = torch.from_numpy(synth.make_waveforms(3, 32000, seed=48)).to(dev)
        feats = hip.compute_feat(x)
        dfe = torch.randn(feats.shape, generator=torch.Generator().manual_seed(1)).to(dev)
        def bwd(t, reuse):
            g = torch.empty_like(t)
            hip.ctx.call("sg_an_logmel_backward", N._ptr(t), 3, 32000, N._ptr(dfe), N._ptr(g), reuse, N.current_stream_ptr(dev))
            return g
        hip.compute_feat(x)
        g1 = bwd(x, 1)
        g0 = bwd(x, 0)
        d = (g1 - g0).abs()
        print("fft%d cache=%d: max |reuse - plain| = %.3e of max %.3e; differing samples %d of %d; equal: %s" % (
            bits, cache, d.max().item(), g0.abs().max().item(), int((d > 0).sum()), d.numel(), torch.equal(g1, g0)))
        if not torch.equal(g1, g0):
            idx = d.flatten().argmax().item()
            b, n = divmod(idx, 32000)
            print("   worst at utterance %d sample %d (frame ~%d)" % (b, n, n // 160))
