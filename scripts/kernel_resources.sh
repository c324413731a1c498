#!/bin/bash
# Register, LDS and occupancy figures of every kernel in a .hip file, one line per kernel (compile only, no GPU).
f=${1:-sph_amd/csrc/sph_sweeps.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage \
  -c -o /dev/null "$f" 2>&1 | python3 -c '
import re, sys
name = None
for line in sys.stdin:
    m = re.search(r"remark: .*?(Function Name|VGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        if name: print(name, row)
        name, row = v, {}
    else:
        row[k.split(" ")[0]] = v
if name: print(name, row)
' | sed 's/_ZN3sph12_GLOBAL__N_1//'
