#!/bin/bash
# Round-6 measurement set (profiles/README.md): run on the GPU box through gpurun, results under gpurun_out/r06/.
out=gpurun_out/r06
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench_line.json 2> $out/bench_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -o bench -- python3 bench.py --no-cpu-baseline --no-shard-points --no-other-configs > $out/bench_line_profiled.json 2> $out/prof_bench.log
for B in 32 16 8; do
    rocprofv3 --kernel-trace --output-format csv -d $out/prof_b$B -o step -- python3 tools/step_profile.py $B 20 > $out/step_b$B.log 2>&1
done
bash tools/pmc_tdnn3.sh $out/pmc
python tools/pmc_tdnn3.py $out/pmc > $out/pmc_tdnn3.json
for B in 64 512; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_an$B -o an -- python3 tools/audionet_profile.py $B > $out/an_profile_b$B.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_feco -o feco -- python3 tools/feco_an_profile.py 64 random > $out/feco_profile.log 2>&1
SG_TUNE=1 SG_SK_TRACE=$out/sk3.bin python tools/layer_bench.py --layers=3 --iters 4 --repeats 1 > /dev/null 2>&1
python tools/sk_trace.py $out/sk3.bin > $out/sk_trace_tdnn3.txt 2>&1
SG_TUNE=1 SG_SK_TRACE=$out/sk5.bin python tools/layer_bench.py --layers=5 --iters 4 --repeats 1 > /dev/null 2>&1
python tools/sk_trace.py $out/sk5.bin > $out/sk_trace_tdnn5.txt 2>&1
rm -f $out/sk3.bin $out/sk5.bin
python tools/batch_sweep.py > $out/batch_sweep.txt 2>&1
python tools/audionet_cnn_bench.py 64 128 512 > $out/audionet_cnn_bench.txt 2>&1
python tools/config_bench.py > $out/config_bench.txt 2>&1
python tools/host_overhead.py > $out/host_overhead.txt 2>&1
python tools/mfcc_variant_probe.py --parity > $out/mfcc_precision.txt 2>&1
find $out -name "*.db" -delete
du -sh $out
