cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04p
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04p/bench -- python3 bench.py --no-cpu-baseline --no-shard-points --no-other-configs > gpurun_out/r04p/bench_line_profiled.json 2> gpurun_out/r04p/bench.err
ls gpurun_out/r04p/bench/*/ | head
bash tools/pmc_tdnn3.sh gpurun_out/r04p/pmc_tdnn3
bash tools/pmc_an_fused.sh gpurun_out/r04p/pmc_an512 512
bash tools/pmc_an_fused.sh gpurun_out/r04p/pmc_an64 64
python tools/pmc_tdnn3.py gpurun_out/r04p/pmc_tdnn3 > gpurun_out/r04p/r04_pmc_tdnn3.json
for b in 512 64; do for k in an_cnn_fwd_kernel an_cnn_bwd_kernel; do python tools/pmc_kernel.py gpurun_out/r04p/pmc_an$b $k > gpurun_out/r04p/r04_pmc_${k}_b$b.json; done; done
grep -h "mfma_busy_fraction\|tcc_hit_rate\|traffic_bytes\|lds_bank\|waiting_on_lds" gpurun_out/r04p/*.json
# keep the merge small: drop the raw counter csvs of the PMC passes except the reduced records
find gpurun_out/r04p -name "*counter_collection.csv" -size +2M -delete
du -sh gpurun_out/r04p
