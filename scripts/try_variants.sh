#!/bin/bash
# Development aid: build libsph_amd.so variants with -D overrides and time each on the GPU box.
# usage (CPU side): scripts/try_variants.sh build "NAME:-DFOO=1" ...   (GPU side): scripts/try_variants.sh run NAME...
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p sph_amd/variants
  for v in "$@"; do
    name=${v%%:*}; defs=${v#*:}
    ( cd sph_amd/csrc && for f in sph_kernels.hip sph_sweeps.hip; do hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include $defs -c $f -o /tmp/var_${f%.hip}.o || exit 1; done &&
      hipcc -O3 -std=c++17 -fPIC -I../../include $defs -c sph_api.cpp -o /tmp/var_api.o &&
      hipcc -shared -fPIC --offload-arch=gfx950 /tmp/var_sph_kernels.o /tmp/var_sph_sweeps.o /tmp/var_api.o -o ../variants/libsph_amd_$name.so ) || exit 1
    echo built $name
  done
else
  for name in "$@"; do
    echo "== $name"
    SPH_AMD_LIB=sph_amd/variants/libsph_amd_$name.so timeout -k 10 120 python3 scripts/gpu_evolve.py 100 5 masked || exit 1
  done
fi
