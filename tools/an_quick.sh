#!/bin/bash
# Iteration aid for the AudioNet front-end: its GPU tests, then the per-kernel averages of the undefended PGD loop at 512 and 64
# utterances (gpurun -- 'bash tools/an_quick.sh').
python -m pytest tests/test_gpu_audionet.py -x -q 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 512 64; do rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f2w_$B -- python3 tools/audionet_profile.py $B > /dev/null 2>&1; python - <<PY
import csv,glob
import os; f=max(glob.glob("gpurun_out/f2w_$B/*/*kernel_stats.csv"), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:6]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
done
