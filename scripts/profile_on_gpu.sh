#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + counters for the bench workload, over EXACTLY the window
# bench.py times by default when the driver calls it (pre-roll 60, --warmup 5, --steps 20).
# Each counter pass is its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
# usage: scripts/profile_on_gpu.sh <tag>      outputs under gpurun_out/prof_<tag>_{stats,fetch,write,sq,mix,ta}
set -e
tag=$1
ARGS="--steps 20 --warmup 5 --preroll 60 --no-cpu-baseline --no-extra-configs"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_${tag}_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -- \
    python3 bench.py $ARGS > gpurun_out/prof_${tag}_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${tag}_fetch -- \
    python3 bench.py $ARGS --no-events > gpurun_out/prof_${tag}_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${tag}_write -- \
    python3 bench.py $ARGS --no-events > gpurun_out/prof_${tag}_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_${tag}_sq -- \
    python3 bench.py $ARGS --no-events > gpurun_out/prof_${tag}_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_${tag}_mix -- \
    python3 bench.py $ARGS --no-events > gpurun_out/prof_${tag}_mix.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_${tag}_ta -- \
    python3 bench.py $ARGS --no-events > gpurun_out/prof_${tag}_ta.log 2>&1
echo profile_done
