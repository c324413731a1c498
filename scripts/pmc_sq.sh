#!/bin/bash
# Development aid: instruction mix and issue utilisation of the sweeps at a chosen phase of the dam break.
# usage: pmc_sq.sh <steps before the measured ones>
pre=${1:-200}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sq_a_$pre -- python3 scripts/gpu_short.py 100 $pre 5 ${2:-mask} > /dev/null 2>&1 || exit 1
timeout -s KILL 100 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/sq_b_$pre -- python3 scripts/gpu_short.py 100 $pre 5 ${2:-mask} > /dev/null 2>&1 || exit 1
timeout -s KILL 100 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum --output-format csv -d gpurun_out/sq_c_$pre -- python3 scripts/gpu_short.py 100 $pre 5 ${2:-mask} > /dev/null 2>&1 || exit 1
python3 scripts/pmc_summary.py gpurun_out/sq_a_$pre gpurun_out/sq_b_$pre gpurun_out/sq_c_$pre | grep -E '==|neigh|dens|force|search'
