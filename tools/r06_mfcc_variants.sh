#!/bin/bash
# MFCC float32 kernel variants: waves per block x occupancy target (VGPR budget), rebuilt on the box, timed in the loop
out=gpurun_out/r06_mfcc; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
for v in "8 4" "8 2" "6 3" "4 4" "4 2"; do
  set -- $v
  rm -f build/obj/k_mfcc.o
  make EXTRA="-DSG_MFCC_F32_WAVES=$1 -DSG_MFCC_F32_OCC=$2" > $out/build_$1_$2.log 2>&1 || { echo "build failed $v"; continue; }
  echo "== waves/block $1, min waves/SIMD $2" | tee -a $out/variants.txt
  python tools/mfcc_variant_probe.py 2>&1 | grep -E "^fft32" | tee -a $out/variants.txt
done
rm -f build/obj/k_mfcc.o; make > $out/build_default.log 2>&1
echo "== default build" | tee -a $out/variants.txt
python tools/mfcc_variant_probe.py --parity 2>&1 | tee -a $out/variants.txt
python -m pytest tests/test_gpu_xv.py -x -q -m gpu 2>&1 | tail -5 | tee -a $out/variants.txt
