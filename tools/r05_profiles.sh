#!/bin/bash
# Round-5 measurement set (profiles/README.md): run on the GPU box through gpurun, results under gpurun_out/r05/.
out=gpurun_out/r05
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench_line.json 2> $out/bench_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -o bench -- python3 bench.py --no-cpu-baseline --no-shard-points --no-other-configs > $out/bench_line_profiled.json 2> $out/prof_bench.log
for B in 32 16 8; do
    rocprofv3 --kernel-trace --output-format csv -d $out/prof_b$B -o step -- python3 tools/step_profile.py $B 20 > $out/step_b$B.log 2>&1
done
bash tools/pmc_tdnn3.sh $out/pmc
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_feco -o feco -- python3 tools/feco_an_profile.py 64 random > $out/feco_profile.log 2>&1
for B in 64 512; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_an$B -o an -- python3 tools/audionet_profile.py $B > $out/an_profile_b$B.log 2>&1
done
SG_TUNE=1 SG_FECO_TRACE=1 python tools/feco_an_profile.py 64 random 2> $out/feco_trace.txt | tail -1 >> $out/feco_trace.txt
python tools/batch_sweep.py > $out/batch_sweep.txt 2>&1
python tools/audionet_cnn_bench.py 64 128 512 > $out/audionet_cnn_bench.txt 2>&1
python tools/config_bench.py > $out/config_bench.txt 2>&1
python tools/host_overhead.py > $out/host_overhead.txt 2>&1
find $out -name "*.db" -delete
du -sh $out
