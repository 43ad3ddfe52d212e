out=gpurun_out/r05c; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $out/tests.txt
python tools/feco_two_cu_ab.py 64 > $out/feco_two_cu_ab.txt 2>&1
python tools/feco_two_cu_ab.py 32 >> $out/feco_two_cu_ab.txt 2>&1
python tools/feco_two_cu_ab.py 8 >> $out/feco_two_cu_ab.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_feco -o feco -- python3 tools/feco_an_profile.py 64 random > $out/feco_profile.log 2>&1
SG_TUNE=1 SG_FECO_TRACE=1 python tools/feco_an_profile.py 64 random 2> $out/feco_trace.txt | tail -1 >> $out/feco_trace.txt
python bench.py > $out/bench_line.json 2> $out/bench_err.log
python tools/config_bench.py > $out/config_bench.txt 2>&1
cat $out/tests.txt; cat $out/feco_two_cu_ab.txt
