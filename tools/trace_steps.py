"""Per-stage times of the fused PGD step from a rocprofv3 --kernel-trace CSV (the 20 launches of one gradient step,
identified by their order: mfcc_fwd ... frames_to_wave), averaged over the steps after the first three.

    python tools/trace_steps.py <B> <kernel_trace.csv> [...]
"""
import csv
import statistics
import sys

NAMES = ['mfcc_fwd', 'cmvn_fwd', 'tdnn1 fwd', 'tdnn2 fwd', 'tdnn3 fwd', 'tdnn4 fwd', 'tdnn5 fwd', 'pool_fwd', 'fc1 fwd', 'tail',
         'fc1 bwd', 'pool_bwd', 'dgrad5', 'dgrad4', 'dgrad3', 'dgrad2', 'dgrad1', 'cmvn_bwd', 'mfcc_bwd', 'overlap_add']
MACS = {'tdnn1': 22732800, 'tdnn2': 377487360, 'tdnn3': 495452160, 'tdnn4': 70778880, 'tdnn5': 207360000}  # per utterance
PEAK = 157.3


def table(B, path):
    rows = [r for r in csv.DictReader(open(path)) if 'sg::' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    seqs, i = [], 0
    while i < len(rows):
        if 'mfcc_fwd' in rows[i]['Kernel_Name'] and i + 19 < len(rows) and 'frames_to_wave' in rows[i + 19]['Kernel_Name']:
            seqs.append(rows[i:i + 20])
            i += 20
        else:
            i += 1
    seqs = seqs[3:]
    print("B=%d: %d steps from %s" % (B, len(seqs), path))
    tot = gemm = flops = 0.0
    for k, n in enumerate(NAMES):
        d = statistics.mean((int(s[k]['End_Timestamp']) - int(s[k]['Start_Timestamp'])) / 1e3 for s in seqs)
        tot += d
        key = 'tdnn' + n[-1] if n.startswith('dgrad') else n.split()[0]
        fl = MACS.get(key)
        extra = ''
        if fl:
            tf = 2 * fl * B / d / 1e6
            extra = ' %6.1f TFLOP/s (%.2f of the f32 MFMA peak)  %s' % (tf, tf / PEAK, seqs[0][k]['Kernel_Name'][10:50])
            if key != 'tdnn1':
                gemm += d
                flops += 2 * fl * B
        print("  %-12s %8.1f us%s" % (n, d, extra))
    print("  sum %.1f us; the 8 stream-K contractions %.1f us = %.1f TFLOP/s (%.3f of peak), the other 12 launches %.1f us"
          % (tot, gemm, flops / gemm / 1e6, flops / gemm / 1e6 / PEAK, tot - gemm))


if __name__ == "__main__":
    args = sys.argv[1:]
    for b, p in zip(args[0::2], args[1::2]):
        table(int(b), p)
