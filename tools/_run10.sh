cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
for B in 64 512; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04f/an$B -- python3 tools/audionet_profile.py $B > gpurun_out/r04f/an$B.log 2>&1
  cp gpurun_out/r04f/an$B/*/*_kernel_stats.csv gpurun_out/r04f/r04_audionet_fused_kernel_stats_b$B.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04f/anfeco -- python3 tools/feco_an_profile.py 64 random > gpurun_out/r04f/anfeco.log 2>&1
cp gpurun_out/r04f/anfeco/*/*_kernel_stats.csv gpurun_out/r04f/r04_audionet_feco_fused_kernel_stats_b64.csv
(python tools/an_trace.py 64; python tools/an_trace.py 512) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04f/r04_an_trace.txt
python tools/audionet_cnn_bench.py 64 128 512 2>&1 | grep -v amdgpu.ids > gpurun_out/r04f/r04_audionet_cnn_bench.txt
python tools/config_bench.py 2>/dev/null > gpurun_out/r04f/r04_config_bench.txt
python bench.py > gpurun_out/r04f/r04_bench_line.json 2> gpurun_out/r04f/bench.err
python tools/batch_sweep.py > gpurun_out/r04f/r04_batch_sweep.txt 2>&1
rm -rf gpurun_out/r04f/an64 gpurun_out/r04f/an512 gpurun_out/r04f/anfeco
tail -3 gpurun_out/r04f/r04_audionet_cnn_bench.txt; head -c 300 gpurun_out/r04f/r04_bench_line.json
mkdir -p gpurun_out/r04p2
bash tools/pmc_an_fused.sh gpurun_out/r04p2/pmc_an512 512
bash tools/pmc_an_fused.sh gpurun_out/r04p2/pmc_an64 64
for b in 512 64; do for k in an_cnn_fwd_kernel an_cnn_bwd_kernel; do python tools/pmc_kernel.py gpurun_out/r04p2/pmc_an$b $k > gpurun_out/r04f/r04_pmc_${k}_b$b.json; done; done
rm -rf gpurun_out/r04p2
grep -h "mfma_busy_fraction" gpurun_out/r04f/r04_pmc_*.json
