#!/bin/bash
# Development aid: rocprofv3 per-kernel durations of a short dam-break run.  usage: kstats.sh <pre-steps> <steps> [mode]
pre=${1:-0}; steps=${2:-20}; mode=${3:-mask}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kstats_$pre
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_$pre -- python3 scripts/gpu_short.py 100 $pre $steps $mode > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/kstats_$pre/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    name = r["Name"].split("(anonymous namespace)::")[-1].split("(")[0]
    print("%-28s calls %5s avg %9.1f us min %9.1f max %9.1f  %5.1f%%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
