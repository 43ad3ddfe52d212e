#!/bin/bash
# PMC passes of the round-6 MFCC kernels inside the PGD loop at 64 utterances (one counter group per pass)
export TMPDIR=/tmp
out=gpurun_out/r06_pmc_mfcc; rm -rf $out /tmp/pmc_mfcc; mkdir -p $out /tmp/pmc_mfcc
bash tools/pmc_run.sh /tmp/pmc_mfcc/b64 tools/step_profile.py 64 20
for k in mfcc_fwd_kernel mfcc_bwd_kernel; do python tools/pmc_kernel.py /tmp/pmc_mfcc/b64 $k > $out/pmc_${k}_b64.json; done
ls -la $out
