#!/bin/bash
# the driver's GPU gate, with the per-test durations table: build, GPU suite, smoke
out=gpurun_out/r06_suite; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
nproc > $out/host.txt
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1; echo "build rc $?"
start=$(date +%s)
python -m pytest tests -m gpu -x -q --durations=0 --durations-min=0.5 > $out/tests_full.txt 2>&1; echo "pytest rc $?"
echo "suite wall $(( $(date +%s) - start )) s" | tee $out/wall.txt
grep -E "passed|failed|error" $out/tests_full.txt | tail -3
sed -n '/slowest/,/^=.*short test summary\|passed/p' $out/tests_full.txt > $out/durations.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
