"""What PGD.attack() costs over the bare fused C-ABI loop on AudioNet (PGD-20 x 64 x 3 s), and where."""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speakerguard_amd import synth
from speakerguard_amd.attack.PGD import PGD
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy
from speakerguard_amd.model.audionet_csine import audionet_csine
dev = torch.device("cuda:0")
m = audionet_csine.from_weights(synth.make_audionet_state_dict(seed=0, num_class=251), device=dev)
x = torch.from_numpy(synth.make_waveforms(64, 48000, seed=5)).to(dev)
y = m.make_decision(x)[0]
lo, hi = torch.clamp(x - 0.002, min=-1), torch.clamp(x + 0.002, max=1)
spec = SEC4SR_CrossEntropy()
K = 20
def t(fn, n=9):
    fn(); fn()
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
    return 1e3 * statistics.median(out), 1e3 * min(out)
a = t(lambda: m.pgd_run(x, y, lo, hi, spec, 0.0004, K, 1))
atk = PGD(m, task="CSI", epsilon=0.002, step_size=0.0004, max_iter=K, batch_size=64, verbose=0)
b = t(lambda: atk.attack(x, y))
print("bare sg_an_pgd_run: median %.3f ms (min %.3f) per PGD-%d; PGD.attack(): median %.3f ms (min %.3f): +%.3f ms" % (a[0], a[1], K, b[0], b[1], b[0] - a[0]))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable(); atk.attack(x, y); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3000])
