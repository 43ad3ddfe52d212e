#!/bin/bash
# rocprofv3 --pmc passes of any workload script, one counter group per pass, --kernel-trace only (never with --sys-trace etc.).
#   gpurun -- 'bash tools/pmc_run.sh gpurun_out/pmc_x tools/audionet_profile.py 512'  then  python tools/pmc_kernel.py gpurun_out/pmc_x KERNEL_SUBSTRING
out=$1
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/$tag" -- python3 "$@" > "$out.$tag.log" 2>&1
done
