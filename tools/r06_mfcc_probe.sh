#!/bin/bash
out=gpurun_out/r06_mfcc; mkdir -p $out; export TMPDIR=/tmp
python tools/mfcc_variant_probe.py --parity 2>&1 | grep -v amdgpu.ids | tee $out/probe_$1.txt
python -m pytest tests/test_gpu_xv.py tests/test_gpu_full_configs.py -x -q -m gpu 2>&1 | tail -5 | tee -a $out/probe_$1.txt
