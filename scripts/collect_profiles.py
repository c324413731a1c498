"""Copies the rocprofv3 summaries of scripts/profile_on_gpu.sh from gpurun_out/ into profiles/ and derives
the per-launch HBM traffic of every kernel over the bench's timed region (the last `steps` launches).

    python scripts/collect_profiles.py <tag> [steps]

FETCH_SIZE / WRITE_SIZE are reported in KB.  On gfx950 FETCH_SIZE counts 64 B per 128-B read request, so it is
doubled (checked here against k_hash, which reads exactly 12 B per particle: x, y, z).
"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

def short(name):
    return name.split("(anonymous namespace)::")[-1].split("(")[0].replace("void ", "")

def newest(pattern):
    # gpurun merges every call's files into gpurun_out/, so earlier runs of the same tag linger: take the latest
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)

stats = newest(os.path.join(src, f"prof_{tag}_stats", "**", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats_whole_run.csv"))
# rocprofv3's own summary averages every launch of the process, pre-roll and rest-density extra included.  The figure
# that bench.py's roofline must agree with is the average over the TIMED WINDOW: the last `steps` launches of each
# kernel in the kernel trace of the same run.  Written in the summary's own format.
trace = newest(os.path.join(src, f"prof_{tag}_stats", "**", "*kernel_trace.csv"))
per_kernel = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    per_kernel[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
window_rows = []
for name, v in per_kernel.items():
    v.sort()
    d = [x[1] for x in v[-steps:]] if len(v) >= steps else [x[1] for x in v]
    if "sph::" not in name or len(v) < steps:
        continue  # one-off kernels (upload/download) are not part of a step
    window_rows.append({"Name": name, "Calls": len(d), "TotalDurationNs": sum(d), "AverageNs": sum(d) / len(d),
                        "MinNs": min(d), "MaxNs": max(d)})
tot = sum(r["TotalDurationNs"] for r in window_rows) or 1
window_rows.sort(key=lambda r: -r["TotalDurationNs"])
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    w.writeheader()
    for r in window_rows:
        r["Percentage"] = 100.0 * r["TotalDurationNs"] / tot
        w.writerow(r)
stats = os.path.join(dst, f"{tag}_kernel_stats.csv")
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

def multi_counters(kind):
    """{kernel: {counter: mean over the timed region}} of a pass that collected several counters."""
    try:
        f = newest(os.path.join(src, f"prof_{tag}_{kind}", "**", "*counter_collection.csv"))
    except ValueError:
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "sph::" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v[-steps:]) / len(v[-steps:]) for c, v in cs.items()} for k, cs in agg.items()}

fetch, write = counters("fetch"), counters("write")
n = bench["config"]["particles"]
out = {"workload": bench["config"]["workload"], "particles": n, "steps_averaged": steps,
       "window": {"preroll": bench["phase"]["preroll_steps"], "warmup": bench["warmup"], "steps": bench["steps"]},
       "units": "bytes per launch, averaged over the timed window; hbm = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction)",
       "kernels": {}}
for k in sorted(fetch):
    out["kernels"][k] = {"fetch_size_kb_raw": fetch[k], "write_size_kb": write.get(k, 0.0),
                         "hbm_bytes_per_launch": (2 * fetch[k] + write.get(k, 0.0)) * 1024.0}
hash_read = 2 * fetch.get("k_hash", 0.0) * 1024.0
out["calibration"] = {"k_hash_read_bytes_corrected": hash_read, "k_hash_read_bytes_expected": 12.0 * n}
force = [k for k in out["kernels"] if k.startswith("k_force")]
out["force_bytes_per_launch"] = out["kernels"][force[0]]["hbm_bytes_per_launch"] if force else None
# Issue side.  SQ counters are summed over the device.  A wave64 vector instruction issues in 2 cycles on a SIMD-32
# at full rate; transcendentals (v_rcp/rsq/sqrt), fp64 arithmetic, conversions and 64-bit integer operations take 4.
# weighted issue cycles / (1024 SIMDs x kernel cycles) is the fraction of the chip's vector issue capacity in use.
# The gathering sweeps are bounded elsewhere: every per-lane 16-byte gather is one vector-cache (TCP) lookup
# (TCP_TOTAL_CACHE_ACCESSES), and a CU's cache serves about one such lookup per cycle.
sq, mix, ta = multi_counters("sq"), multi_counters("mix"), multi_counters("ta")
out["issue"] = {}
for k, c in sq.items():
    if "SQ_INSTS_VALU" not in c:
        continue
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the 8 XCDs
    row = {"valu_wave_insts_per_launch": c["SQ_INSTS_VALU"],
           "lane_utilisation": c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * c["SQ_INSTS_VALU"]),
           "waves_per_simd": 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / (1024.0 * cyc) if cyc else None,
           "wave_time_waiting_frac": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None}
    m = mix.get(k, {})
    if m.get("SQ_INSTS_VALU") and m.get("GRBM_GUI_ACTIVE"):
        slow = sum(m.get(x, 0.0) for x in ("SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
                                           "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT64"))
        weighted = 2.0 * m["SQ_INSTS_VALU"] + 2.0 * slow
        row["valu_half_rate_insts_per_launch"] = slow
        row["valu_issue_cycles_per_launch"] = weighted
        row["valu_issue_frac"] = weighted / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
        row["valu_issue_cycles_per_particle_lane"] = weighted * 64.0 / n
    t = ta.get(k, {})
    if t.get("GRBM_GUI_ACTIVE"):
        tcyc = t["GRBM_GUI_ACTIVE"] / 8.0
        row["ta_busy_frac"] = t.get("TA_TA_BUSY_sum", 0.0) / 256.0 / tcyc
        row["tcp_lane_accesses_per_launch"] = t.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)
        row["tcp_accesses_per_cu_cycle"] = t.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / 256.0 / tcyc
        if t.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            row["l1_hit_frac"] = 1.0 - t.get("TCP_TCC_READ_REQ_sum", 0.0) / t["TCP_TOTAL_CACHE_ACCESSES_sum"]
    out["issue"][k] = row
out["issue_side"] = {k: {a: v for a, v in row.items() if a in ("valu_issue_frac", "valu_issue_cycles_per_particle_lane",
                                                               "lane_utilisation", "ta_busy_frac",
                                                               "tcp_accesses_per_cu_cycle", "l1_hit_frac")}
                     for k, row in out["issue"].items() if k.startswith(("k_force", "k_neighbours", "k_density"))}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
rows = list(csv.DictReader(open(stats)))
print(f"{tag}: bench under rocprof: {bench['value']:.3e} pu/s, {bench['ms_per_step']:.3f} ms/step")
for r in rows[:10]:
    print("  %-22s calls %5s avg %9.1f us  %5.1f%%" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("  calibration k_hash read: %.1f MB corrected vs %.1f MB expected" % (hash_read / 1e6, 12.0 * n / 1e6))
for k, v in out["kernels"].items():
    print("  %-22s hbm %8.1f MB/launch" % (k, v["hbm_bytes_per_launch"] / 1e6))
for k, v in out["issue"].items():
    print("  %-22s" % k, " ".join("%s=%.3g" % (a, b) for a, b in v.items() if b is not None))
