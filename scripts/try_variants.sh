#!/bin/bash
# Development aid: build libsph_amd.so variants with -D overrides (both kernel files, the Makefile's flags) and time
# each on the GPU box.
# usage (CPU side): scripts/try_variants.sh build "NAME:-DFOO=1" ...
#       (GPU side): scripts/try_variants.sh run <scene> <side> <first_step> <steps> NAME...
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p sph_amd/variants
  make -C sph_amd/csrc -j8 > /dev/null || exit 1
  for v in "$@"; do
    name=${v%%:*}; defs=${v#*:}
    ( cd sph_amd/csrc &&
      hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 $defs -c sph_sweeps.hip -o /tmp/var_sweeps_$name.o &&
      hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 $defs -c sph_kernels.hip -o /tmp/var_kernels_$name.o &&
      hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libsph_amd_$name.so /tmp/var_kernels_$name.o /tmp/var_sweeps_$name.o sph_render.o sph_api.o sph_rccl.o -ldl ) || exit 1
    echo built $name
  done
else
  scene=$1; side=$2; first=$3; steps=$4; shift 4
  for name in "$@"; do
    echo "== $name"
    SPH_AMD_LIB=sph_amd/variants/libsph_amd_$name.so timeout -k 10 300 python3 scripts/gpu_window_kernels.py $scene $side $first $steps || exit 1
  done
fi
