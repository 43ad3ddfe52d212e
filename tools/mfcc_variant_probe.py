"""MFCC forward / adjoint stage times inside a traced PGD attack (64 and 8 utterances x 3 s) and the parity numbers of the
float32 / float64 transforms against the oracle: cepstra, d loss / d wav (max error, sign mismatches).  GPU box only."""
import os, statistics, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(16)
from speakerguard_amd import synth
from speakerguard_amd.attack.PGD import PGD
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.xv_plda import xv_plda
dev = torch.device("cuda:0")
w = synth.make_xv_weights()
m = xv_plda.from_weights(w, device=dev, dither=0.0)
parity = "--parity" in sys.argv
for bits in (32, 64):
    m.configure_frontend(bits)
    for B in (64, 8):
        x = torch.from_numpy(synth.make_waveforms(B, 48000, seed=1234)).to(dev)
        y = (torch.arange(B) % 10).to(dev)
        atk = PGD(m, task="CSI", epsilon=0.002, step_size=0.0004, max_iter=20, batch_size=B, verbose=0)
        atk.attack(x, y)
        recs = m.trace_stages(lambda: atk.attack(x, y), max_records=64 * 22)
        by = {}
        for k, ms in recs:
            by.setdefault(k, []).append(ms)
        tot = sum(sum(v) for v in by.values()) / 20
        print("fft%d B=%d: mfcc_fwd %.1f us  mfcc_bwd %.1f us  overlap_add %.1f us  (all stages %.1f us/step)" % (
            bits, B, 1e3 * statistics.mean(by["mfcc_fwd"]), 1e3 * statistics.mean(by["mfcc_bwd"]),
            1e3 * statistics.mean(by["overlap_add"]), 1e3 * tot))
if parity:
    from oracle import kaldi_mfcc, attacks as oatk
    from oracle.xv_plda import XvPlda
    om = XvPlda(w, faithful=False)
    x = torch.from_numpy(synth.make_waveforms(3, 48000, seed=22))
    with torch.no_grad():
        y = om.make_decision(x)[0]
    xin = x.clone().requires_grad_(True)
    oatk.cross_entropy_loss(om.make_decision(xin)[1], y).backward(torch.ones(3))
    want = xin.grad.numpy()
    m64 = XvPlda(w).double()
    x64 = x.double().requires_grad_(True)
    torch.nn.functional.cross_entropy(m64.make_decision(x64)[1], y, reduction="none").backward(torch.ones(3, dtype=torch.float64))
    g64 = x64.grad.numpy()
    wantc = kaldi_mfcc.mfcc_batch(x * 32768.0).numpy()
    c64 = kaldi_mfcc.mfcc_batch(x.double() * 32768.0).numpy() if hasattr(kaldi_mfcc, "mfcc_batch") else None
    def stats(nm, arr, ref):
        return "%s: max err/max %.3e rms/rms %.3e sign mismatch %.3e" % (nm, np.abs(arr - ref).max() / np.abs(ref).max(),
                np.sqrt(((arr - ref) ** 2).mean() / (ref ** 2).mean()), float((np.sign(arr) != np.sign(ref)).mean()))
    print(stats("oracle-fp32 grad vs fp64 truth", want, g64))
    for bits in (32, 64):
        m.configure_frontend(bits)
        got = m.loss_grad(x.to(dev), y.to(dev), SEC4SR_CrossEntropy())[3].cpu().numpy()
        c = m.compute_feat(x.to(dev), flag=1).cpu().numpy()
        print("fft%d cepstra vs oracle-fp32: max abs err %.3e" % (bits, np.abs(c - wantc).max()))
        try:
            print("fft%d cepstra vs oracle-fp64: max abs err %.3e (oracle-fp32 vs fp64: %.3e)" % (bits, np.abs(c - c64).max(), np.abs(wantc - c64).max()))
        except Exception as e:
            print("no fp64 cepstra:", e)
        print("fft%d " % bits + stats("hip grad vs oracle-fp32", got, want))
        print("fft%d " % bits + stats("hip grad vs fp64 truth", got, g64))
