#!/bin/bash
# Headless multi-slab driver on the GPU: its tests, then a 1M-particle dam break on 1 context and on 4 slabs.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_shim.py -x -v --timeout 300 --timeout-method thread -m gpu -k "headless" \
  > gpurun_out/headless_slabs_tests.log 2>&1 || { tail -40 gpurun_out/headless_slabs_tests.log; exit 1; }
tail -12 gpurun_out/headless_slabs_tests.log
cd gpurun_out &&
timeout -k 10 200 ../sph_amd/sph_headless --scene dam --side 100 --steps 100 --mode resident --out h1 > headless_1m_one.json &&
timeout -k 10 200 ../sph_amd/sph_headless --scene dam --side 100 --steps 100 --mode resident --slabs 4 --out h4 > headless_1m_slabs4.json &&
cat headless_1m_one.json headless_1m_slabs4.json && cmp h1_final.sphd h4_final.sphd && echo "dumps identical" && rm -f h1_final.sphd h4_final.sphd
