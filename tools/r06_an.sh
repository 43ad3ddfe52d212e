#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06_an; rm -rf $out /tmp/pmc_an; mkdir -p $out /tmp/pmc_an
python -m pytest tests/test_gpu_audionet.py tests/test_gpu_feco.py tests/test_gpu_full_configs.py -x -q -m gpu 2>&1 | tail -3 | tee $out/tests.txt
python tools/an_head_ab.py 2>&1 | grep "B=" | tee $out/head_ab.txt
python tools/an_head_loop_ab.py 2>&1 | grep "B=" | tee $out/head_loop_ab.txt
for B in 64 512; do
  bash tools/pmc_run.sh /tmp/pmc_an/b$B tools/audionet_profile.py $B
  for k in an_cnn_fwd_kernel an_cnn_bwd_kernel; do python tools/pmc_kernel.py /tmp/pmc_an/b$B $k > $out/pmc_${k}_b$B.json; done
done
ls $out
