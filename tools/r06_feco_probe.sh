#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_gpu_feco.py -x -q -m gpu 2>&1 | tail -2
SG_TUNE=1 SG_FECO_TRACE=1 python tools/feco_an_profile.py 64 random 2>&1 | grep -E "phases|ms per step" | sed -n '3,6p;$p'
python tools/feco_two_cu_ab.py 64 2>&1 | tail -3
