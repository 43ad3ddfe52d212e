"""Per-layer timing of the TDNN contractions (forward and data-gradient) on one GPU.

    python tools/layer_bench.py [--batch 64] [--iters 0 --round-ms 15]

Prints ms / TFLOP/s / fraction of the 157.3 TFLOP/s f32-MFMA peak per launch.  Used for kernel
tuning and as the workload of the rocprofv3 --pmc passes kept under profiles/.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from speakerguard_amd import synth  # noqa: E402
from speakerguard_amd.attack.utils import SEC4SR_CrossEntropy  # noqa: E402
from speakerguard_amd.model.xv_plda import xv_plda  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=0,
                    help="launches per timing round; 0 (default) = as many as fill --round-ms (at least 20)")
    ap.add_argument("--round-ms", type=float, default=15.0,
                    help="length of a timing round when --iters is 0: the chip needs several milliseconds of continuous work to "
                         "reach its clock (20 launches of a 70 us kernel read 15 %% slow, profiles/r03_staging_cost.txt)")
    ap.add_argument("--layers", type=str, default="1,2,3,4,5,-5,-4,-3,-2,-1")
    ap.add_argument("--repeats", type=int, default=5, help="timing rounds per layer; the best and the median are printed")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    model = xv_plda.from_weights(synth.make_xv_weights(), device=dev, dither=0.0)
    x = torch.from_numpy(synth.make_waveforms(args.batch, 48000, seed=1234)).to(dev)
    y = (torch.arange(args.batch) % 10).to(dev)
    model.loss_grad(x, y, SEC4SR_CrossEntropy())  # fills activations and gradients
    torch.cuda.synchronize()
    tot_ms, tot_fl = 0.0, 0.0
    for l in [int(v) for v in args.layers.split(",")]:
        iters = args.iters
        if iters <= 0:
            est = model.time_layer(l, args.batch, 48000, 20)[0]
            iters = max(20, int(args.round_ms / max(est, 1e-3)))
        runs = sorted(model.time_layer(l, args.batch, 48000, iters) for _ in range(args.repeats))
        ms, fl, rows = runs[0]
        med = runs[len(runs) // 2][0]
        tf = fl / (ms * 1e-3) / 1e12
        tot_ms += ms
        tot_fl += fl
        print("layer %+d  tile_rows %3d  %8.3f ms  %7.2f TFLOP/s  %5.1f %% of 157.3   (median %.3f ms)" % (
            l, rows, ms, tf, 100 * tf / 157.3, med))
    print("sum        %8.3f ms  %7.2f TFLOP/s" % (tot_ms, tot_fl / (tot_ms * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
