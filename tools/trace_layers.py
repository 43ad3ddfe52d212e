"""Per-LAYER kernel times from a rocprofv3 kernel trace of ``bench.py``.

``rocprofv3 --kernel-trace --stats`` averages ``conv_gemm_streamk_kernel<1, 2>`` over the four forward layers that
share that instantiation (and ``<2, 2>`` over the four data-gradient layers), so its per-kernel average cannot be
compared with ``roofline.ms_per_launch`` (tdnn3 forward alone).  This tool walks the dispatches of the SAME trace in
time order and names every contraction by its position in the pass:

    forward pass :  tdnn1 (conv_gemm_q_kernel<1, *>)  ->  the next four streamk<1, *> launches = tdnn2..tdnn5
    backward pass:  tail_kernel                        ->  the next four streamk<2, *> launches = tdnn5..tdnn2

Launches outside such a pass (``sg_xv_time_layer``: the HIP-event timing of tdnn3 forward that feeds ``roofline``) are
reported as their own row, so the profiler's view and the library's HIP-event view of the same launches sit side by side.

    python tools/trace_layers.py <dir or *_kernel_trace.csv> [bench line .json]
"""
import csv
import glob
import json
import os
import sys

FLOP = {  # per launch at B = 64, 3 s: 2 * M * N * K
    "tdnn2 fwd": 2 * 64 * 288 * 512 * 2560, "tdnn3 fwd": 2 * 64 * 270 * 512 * 3584, "tdnn4 fwd": 2 * 64 * 270 * 512 * 512,
    "tdnn5 fwd": 2 * 64 * 270 * 1536 * 512,
    "tdnn5 dgrad": 2 * 64 * 270 * 512 * 1536, "tdnn4 dgrad": 2 * 64 * 270 * 512 * 512,
    "tdnn3 dgrad": 2 * 64 * 288 * 512 * 3584, "tdnn2 dgrad": 2 * 64 * 296 * 512 * 2560,
}


def find_trace(path):
    if os.path.isfile(path):
        return path
    hits = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    if not hits:
        sys.exit("no *kernel_trace.csv under %s" % path)
    return hits[-1]


def main():
    trace = find_trace(sys.argv[1])
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    fwd_left, bwd_left = 0, 0
    acc = {}
    for r in rows:
        name = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "conv_gemm_q_kernel<1," in name:
            fwd_left, bwd_left = 4, 0
            continue
        if "tail_kernel" in name and "an_tail" not in name:
            bwd_left, fwd_left = 4, 0
            continue
        if "conv_gemm_streamk_kernel<1," in name:
            if fwd_left:
                label = "tdnn%d fwd" % (6 - fwd_left)
                fwd_left -= 1
            else:
                label = "sg_xv_time_layer launches (tdnn3 fwd)"
            acc.setdefault(label, []).append(dur)
        elif "conv_gemm_streamk_kernel<2," in name:
            if bwd_left:
                label = "tdnn%d dgrad" % (bwd_left + 1)
                bwd_left -= 1
            else:
                label = "unattributed streamk<2>"
            acc.setdefault(label, []).append(dur)
    print("per-layer stream-K launch times from %s" % os.path.basename(trace))
    print("%-40s %6s %10s %10s %10s %9s" % ("layer", "calls", "avg us", "min us", "max us", "TFLOP/s"))
    order = ["tdnn2 fwd", "tdnn3 fwd", "tdnn4 fwd", "tdnn5 fwd", "tdnn5 dgrad", "tdnn4 dgrad", "tdnn3 dgrad", "tdnn2 dgrad",
             "sg_xv_time_layer launches (tdnn3 fwd)", "unattributed streamk<2>"]
    for k in order:
        if k not in acc:
            continue
        v = acc[k]
        avg = sum(v) / len(v)
        fl = FLOP.get(k, FLOP["tdnn3 fwd"] if "time_layer" in k else None)
        print("%-40s %6d %10.1f %10.1f %10.1f %9s" % (k, len(v), avg, min(v), max(v), "%.1f" % (fl / avg / 1e6) if fl else "-"))
    if len(sys.argv) > 2:
        line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
        ms = line["roofline"]["ms_per_launch"]
        print("bench line of the same run: roofline.ms_per_launch = %.1f us (HIP events inside the library), frac %.3f"
              % (ms * 1e3, line["roofline"]["frac"]))
        for k in ("tdnn3 fwd", "sg_xv_time_layer launches (tdnn3 fwd)"):
            if k in acc:
                avg = sum(acc[k]) / len(acc[k])
                print("  profiler average of '%s': %.1f us (%+.1f %% against the HIP-event figure)" % (k, avg, 100 * (avg / (ms * 1e3) - 1)))


if __name__ == "__main__":
    main()
