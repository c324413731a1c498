"""Copies the rocprofv3 summaries of scripts/profile_on_gpu.sh from gpurun_out/ into profiles/ and derives
the per-launch HBM traffic of every kernel over the bench's timed region (the last `steps` launches).

    python scripts/collect_profiles.py <tag> [steps]

FETCH_SIZE / WRITE_SIZE are reported in KB.  On gfx950 FETCH_SIZE counts 64 B per 128-B read request, so it is
doubled (checked here against k_hash, which reads exactly 12 B per particle: x, y, z).
"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

def short(name):
    return name.split("(anonymous namespace)::")[-1].split("(")[0].replace("void ", "")

def newest(pattern):
    # gpurun merges every call's files into gpurun_out/, so earlier runs of the same tag linger: take the latest
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)

stats = newest(os.path.join(src, f"prof_{tag}_stats", "**", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench_line = [l for l in open(os.path.join(src, f"prof_{tag}_stats.log")) if l.startswith("{")][-1]
open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(bench_line)
bench = json.loads(bench_line)

def counters(kind):
    f = newest(os.path.join(src, f"prof_{tag}_{kind}", "**", "*counter_collection.csv"))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sph::" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v[-steps:]) / len(v[-steps:]) for k, v in agg.items()}

fetch, write = counters("fetch"), counters("write")
n = bench["config"]["particles"]
out = {"workload": bench["config"]["workload"], "particles": n, "steps_averaged": steps,
       "units": "bytes per launch, averaged over the timed region; hbm = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction)",
       "kernels": {}}
for k in sorted(fetch):
    out["kernels"][k] = {"fetch_size_kb_raw": fetch[k], "write_size_kb": write.get(k, 0.0),
                         "hbm_bytes_per_launch": (2 * fetch[k] + write.get(k, 0.0)) * 1024.0}
hash_read = 2 * fetch.get("k_hash", 0.0) * 1024.0
out["calibration"] = {"k_hash_read_bytes_corrected": hash_read, "k_hash_read_bytes_expected": 12.0 * n}
force = [k for k in out["kernels"] if k.startswith("k_force")]
out["force_bytes_per_launch"] = out["kernels"][force[0]]["hbm_bytes_per_launch"] if force else None
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
rows = list(csv.DictReader(open(stats)))
print(f"{tag}: bench under rocprof: {bench['value']:.3e} pu/s, {bench['ms_per_step']:.3f} ms/step")
for r in rows[:10]:
    print("  %-22s calls %5s avg %9.1f us  %5.1f%%" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("  calibration k_hash read: %.1f MB corrected vs %.1f MB expected" % (hash_read / 1e6, 12.0 * n / 1e6))
for k, v in out["kernels"].items():
    print("  %-22s hbm %8.1f MB/launch" % (k, v["hbm_bytes_per_launch"] / 1e6))
