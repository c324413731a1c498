#!/bin/bash
# Development aid: vector / scalar instruction counts and busy cycles of the sweeps at a chosen phase of the dam break.
# usage: pmc_valu.sh <steps before the measured ones> [side]
pre=${1:-65}; side=${2:-100}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/valu_$pre -- python3 scripts/gpu_short.py $side $pre 5 mask > /dev/null 2>&1 || exit 1
python3 scripts/pmc_summary.py gpurun_out/valu_$pre | grep -E '==|neigh|force'
