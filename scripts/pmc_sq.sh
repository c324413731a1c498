#!/bin/bash
# Development aid: instruction mix and issue utilisation of the sweeps in the dense phase of the dam break.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sq_a -- python3 scripts/gpu_short.py 100 200 5 mask > /dev/null 2>&1 || exit 1
timeout -s KILL 100 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/sq_b -- python3 scripts/gpu_short.py 100 200 5 mask > /dev/null 2>&1 || exit 1
python3 scripts/pmc_summary.py gpurun_out/sq_a gpurun_out/sq_b | grep -E '==|neigh|dens|force'
