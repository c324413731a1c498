#!/bin/bash
# Development aid (GPU box): average duration of every sph kernel over the first and the last 20 launches of a run of
# the 1M dam break to step 85 (rest density / peak compression).  usage: kernel_window.sh [label]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kwin
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kwin -- python3 scripts/gpu_short.py 100 65 20 ${2:-mask} > /dev/null 2>&1 || { echo "run failed"; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/kwin/**/*kernel_trace.csv", recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]
    d.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("${1:-run}:", "  ".join("%s %.1f/%.1f" % (k.replace("k_", ""), sum(v[:20]) / 20e3, sum(v[-20:]) / 20e3) for k, v in d.items() if len(v) >= 40))
PY
