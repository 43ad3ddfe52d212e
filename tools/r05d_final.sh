#!/bin/bash
# the round's closing check on a fresh box: build, GPU suite, smoke, bench line (what the driver runs at round end)
out=gpurun_out/r05d; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1; echo "build rc $?" 
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $out/tests.txt; cat $out/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $out/bench_line.json 2> $out/bench_err.log; tail -c 600 $out/bench_line.json
python tools/config_bench.py > $out/config_bench.txt 2>&1
