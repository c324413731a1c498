#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + HBM traffic counters for the bench workload.
# Each counter pass is its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
# usage: scripts/profile_on_gpu.sh <tag>      outputs under gpurun_out/prof_<tag>_{stats,fetch,write}
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -- \
    python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/prof_${tag}_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${tag}_fetch -- \
    python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-events > gpurun_out/prof_${tag}_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${tag}_write -- \
    python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-events > gpurun_out/prof_${tag}_write.log 2>&1
echo profile_done
