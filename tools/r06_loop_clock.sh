#!/bin/bash
# The shader clock and the chunk times of the stream-K launches INSIDE the headline loop (not in a 4-launch measurement run):
# SG_SK_TRACE with a window (launches 400 .. 415 of the process = two PGD steps after ~50 steps of warm loop).
out=gpurun_out/r06c; mkdir -p $out; rm -f $out/sk_loop.bin
SG_TUNE=1 SG_SK_TRACE=$out/sk_loop.bin SG_SK_TRACE_SKIP=400 SG_SK_TRACE_COUNT=16 timeout 600 python bench.py --steps 100 --warmup 20 > $out/bench_traced.json 2> $out/bench_traced.err
python tools/sk_trace.py $out/sk_loop.bin --all > $out/sk_loop.txt 2>&1
cat $out/sk_loop.txt
# the same launches in a short measurement run, for comparison
rm -f $out/sk_short.bin
SG_TUNE=1 SG_SK_TRACE=$out/sk_short.bin timeout 120 python tools/layer_bench.py --layers=3 --iters 4 --repeats 1 > /dev/null 2>&1
python tools/sk_trace.py $out/sk_short.bin --all 2>&1 | tail -3 | tee $out/sk_short.txt
