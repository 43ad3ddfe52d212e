#!/bin/bash
# Round-5 measurement set, second half (after the AudioNet front-end rewrite): gpurun -- 'bash tools/r05b_profiles.sh'
out=gpurun_out/r05b
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench_line.json 2> $out/bench_err.log
for B in 64 512; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_an$B -o an -- python3 tools/audionet_profile.py $B > $out/an_profile_b$B.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_feco -o feco -- python3 tools/feco_an_profile.py 64 random > $out/feco_profile.log 2>&1
python tools/an_frontend_ab.py 64,0,0 64,1,0 64,0,1 32,0,0 32,1,0 32,0,1 32,1,1 32,1,-1 64 128 256 512 > $out/an_frontend_ab.txt 2>&1
python tools/config_bench.py > $out/config_bench.txt 2>&1
bash tools/pmc_run.sh $out/pmc_an512 tools/audionet_profile.py 512
bash tools/pmc_run.sh $out/pmc_an64 tools/audionet_profile.py 64
find $out -name "*.db" -delete
du -sh $out
