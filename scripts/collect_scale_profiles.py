"""Summarises the outputs of scripts/profile_scale_on_gpu.sh into profiles/<tag>_scale.json: per configuration and
kernel the average duration over the timed window (the last `steps` launches), HBM-side bytes per launch
(2 * FETCH_SIZE + WRITE_SIZE, KB units, the gfx950 correction of collect_profiles.py), vector-cache hit fraction and
texture-addresser busy fraction.    python scripts/collect_scale_profiles.py <tag> [steps]"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
src = os.path.join(ROOT, "gpurun_out")


def short(name):
    return name.split("(anonymous namespace)::")[-1].split("(")[0].replace("void ", "")


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def counter_means(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "sph::" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the window's launches come before the one extra debug step: drop the last launch, average the `steps` before it
    return {k: {c: sum(v[-steps - 1:-1]) / len(v[-steps - 1:-1]) for c, v in cs.items() if len(v) > steps}
            for k, cs in agg.items()}


out = {}
for cfg in ("dam100", "dam200", "cube256"):
    base = os.path.join(src, f"scale_{tag}_{cfg}")
    try:
        line = [l for l in open(base + "_stats.log") if l.startswith("{")][-1]
    except (OSError, IndexError):
        continue
    info = json.loads(line)
    n = info["particles"]
    trace = newest(os.path.join(base + "_stats", "**", "*kernel_trace.csv"))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "sph::" in r["Kernel_Name"]:
            dur[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    fetch = counter_means(newest(os.path.join(base + "_fetch", "**", "*counter_collection.csv")))
    write = counter_means(newest(os.path.join(base + "_write", "**", "*counter_collection.csv")))
    ta = counter_means(newest(os.path.join(base + "_ta", "**", "*counter_collection.csv")))
    kernels = {}
    for k, v in dur.items():
        v.sort()
        if len(v) <= steps:
            continue
        d = [x[1] for x in v[-steps - 1:-1]]
        row = {"avg_us": sum(d) / len(d) / 1e3, "us_per_million_particles": sum(d) / len(d) / 1e3 / (n / 1e6)}
        if k in fetch and "FETCH_SIZE" in fetch[k]:
            hbm = (2 * fetch[k]["FETCH_SIZE"] + write.get(k, {}).get("WRITE_SIZE", 0.0)) * 1024.0
            row["hbm_bytes_per_launch"] = hbm
            row["hbm_bytes_per_particle"] = hbm / n
        t = ta.get(k, {})
        if t.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            row["l1_hit_frac"] = 1.0 - t.get("TCP_TCC_READ_REQ_sum", 0.0) / t["TCP_TOTAL_CACHE_ACCESSES_sum"]
            if t.get("GRBM_GUI_ACTIVE"):
                row["ta_busy_frac"] = t.get("TA_TA_BUSY_sum", 0.0) / 256.0 / (t["GRBM_GUI_ACTIVE"] / 8.0)
        kernels[k] = row
    out[cfg] = {"run": info, "kernels": kernels}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_scale.json"), "w"), indent=1)
for cfg, v in out.items():
    print(cfg, "mean neighbours %.1f" % v["run"]["mean_neighbours"], "ms/step %.3f" % v["run"]["ms_per_step"])
    for k, r in sorted(v["kernels"].items(), key=lambda kv: -kv[1]["avg_us"]):
        print("   %-26s %9.1f us  %7.1f us/M  hbm %6.1f B/particle  L1 hit %.2f  TA busy %.2f" % (
            k, r["avg_us"], r["us_per_million_particles"], r.get("hbm_bytes_per_particle", float("nan")),
            r.get("l1_hit_frac", float("nan")), r.get("ta_busy_frac", float("nan"))))
