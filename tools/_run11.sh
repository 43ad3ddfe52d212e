cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --durations=25 2>&1 | tail -45 > gpurun_out/r04_full_gpu2.log; cat gpurun_out/r04_full_gpu2.log
