cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_full_configs.py tests/test_gpu_multiproc.py tests/test_gpu_post.py tests/test_gpu_xv.py -q -m gpu 2>&1 | tail -12 > gpurun_out/r04_rest_gpu.log; cat gpurun_out/r04_rest_gpu.log
python tools/host_overhead.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_host_overhead.txt; head -30 gpurun_out/r04_host_overhead.txt
