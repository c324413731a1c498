#!/bin/bash
# Runs on the GPU box (via gpurun): how the sweeps' cost per particle depends on the size of the grid.  For the 1M and
# 8M dam break in the same phase of the collapse and the 16.8M uniform cube: kernel trace, FETCH_SIZE, WRITE_SIZE and
# vector-cache counters, each counter pass a run of its own.
# usage: scripts/profile_scale_on_gpu.sh <tag>     outputs under gpurun_out/scale_<tag>_<config>_{stats,fetch,write,ta}
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # name, script arguments
  name=$1; shift
  for pass in stats fetch write ta; do
    case $pass in
      stats) opts="--kernel-trace --stats" ;;
      fetch) opts="--pmc FETCH_SIZE" ;;
      write) opts="--pmc WRITE_SIZE" ;;
      ta)    opts="--pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" ;;
    esac
    out=gpurun_out/scale_${tag}_${name}_${pass}
    rm -rf $out
    timeout -s KILL 280 rocprofv3 $opts --output-format csv -d $out -- python3 scripts/gpu_window_kernels.py "$@" > $out.log 2>&1 || { echo "$name $pass failed"; tail -5 $out.log; return 1; }
    echo "$name $pass done"
  done
}
run dam100 dam_break 100 65 20 && run dam200 dam_break 200 92 20 && run cube256 uniform_cube 256 20 20 && echo scale_profile_done
