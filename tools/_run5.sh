cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_audionet.py -x -q 2>&1 | tail -25 > gpurun_out/r04_t5.log; cat gpurun_out/r04_t5.log
python tests/tools/probe_r04.py 2>&1 | grep -v "^Running\|Early" | tail -40 > gpurun_out/r04_probe.log; cat gpurun_out/r04_probe.log
