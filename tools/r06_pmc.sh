#!/bin/bash
# counters for the data gradients and the K = 512 launches, next to tdnn3 forward from the same box.
# One rocprofv3 --pmc pass per counter group (tools/pmc_run.sh), reduced on the box to one JSON per kernel instantiation.
export TMPDIR=/tmp
out=gpurun_out/r06_pmc; rm -rf $out /tmp/pmc; mkdir -p $out /tmp/pmc
for spec in "3:<1, 8>:tdnn3_fwd" "-3:<2, 8>:dgrad3" "5:<1, 5>:tdnn5_fwd" "-5:<2, 5>:dgrad5"; do
  IFS=: read layer inst name <<< "$spec"
  bash tools/pmc_run.sh /tmp/pmc/$name tools/layer_bench.py --layers=$layer --iters 6 --repeats 1
  python tools/pmc_kernel.py /tmp/pmc/$name "conv_gemm_streamk_kernel$inst" > $out/pmc_$name.json
  tail -3 /tmp/pmc/$name.FETCH_SIZE.log > $out/log_$name.txt
done
ls -la $out
