#!/bin/bash
# rocprofv3 --pmc passes of the dominant kernel (tdnn3 forward contraction, B = 64), one counter group per pass
# (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2; never combined with --sys-trace etc.).
#   gpurun -- 'bash tools/pmc_tdnn3.sh gpurun_out/pmc_r02'   then   python tools/pmc_tdnn3.py gpurun_out/pmc_r02 > profiles/r02_pmc_tdnn3.json
out=${1:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/$tag" -- python3 tools/layer_bench.py --layers 3 --iters 4 --repeats 1 > "$out.$tag.log" 2>&1
done
