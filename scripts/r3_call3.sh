#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > $o/r3_noalloc.log 2>&1; echo "noalloc rc=$?"; tail -3 $o/r3_noalloc.log
timeout -k 10 400 python -u -m pytest tests/test_gpu_slab.py -x -q -s --timeout 300 --timeout-method thread -k "swinging" > $o/r3_slosh_long.log 2>&1; echo "slosh rc=$?"; tail -4 $o/r3_slosh_long.log | cut -c1-1500
timeout -k 10 300 python bench.py --gpus 1 --force-slab --side 100 --steps 10 --warmup 2 --no-cpu-baseline > $o/reh_lib1.json 2> $o/reh_lib1.err; echo "lib1 rc=$?"; tail -c 600 $o/reh_lib1.json
timeout -k 10 300 python scripts/gpu_window_kernels.py dam_break 400 120 20 > $o/r3_64m_single.json 2> $o/r3_64m_single.err; echo "64m single rc=$?"; cut -c1-700 $o/r3_64m_single.json
timeout -k 10 400 python scripts/gpu_balance_work.py 400 8 140 work > $o/r3_64m_slabs.json 2> $o/r3_64m_slabs.err; echo "64m slabs rc=$?"; cut -c1-1200 $o/r3_64m_slabs.json
