#!/bin/bash
# GPU box: rehearsal of the multi-GPU bench legs on one GPU and the single-GPU line with its extra configurations.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
o=gpurun_out
timeout -k 10 300 python bench.py --gpus 2 --side 60 --steps 10 --warmup 2 > $o/reh_plain2.json 2> $o/reh_plain2.err; echo "plain2 rc=$?"; tail -c 600 $o/reh_plain2.json
timeout -k 10 300 python bench.py --gpus 2 --side 60 --steps 10 --warmup 2 --workload sloshing --preroll 10 --no-cpu-baseline > $o/reh_slosh2.json 2> $o/reh_slosh2.err; echo "slosh2 rc=$?"; tail -c 300 $o/reh_slosh2.json
timeout -k 10 300 python bench.py --gpus 2 --side 60 --steps 10 --warmup 2 --workload uniform_cube --no-cpu-baseline > $o/reh_cube2.json 2> $o/reh_cube2.err; echo "cube2 rc=$?"; tail -c 300 $o/reh_cube2.json
timeout -k 10 300 python bench.py --gpus 1 --force-slab --side 100 --steps 10 --warmup 2 --no-cpu-baseline > $o/reh_lib1.json 2> $o/reh_lib1.err; echo "lib1 rc=$?"; tail -c 400 $o/reh_lib1.json
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $o/reh_single.json 2> $o/reh_single.err; echo "single rc=$?"
