"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel over its LAST launches."""
import csv, collections, glob, sys
def main(dirs, last=5):
    for d in dirs:
        for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if "sph::" not in name: continue
                short = name.split("(anonymous namespace)::")[-1].split("(")[0]
                agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
            print("==", d)
            for k, cs in agg.items():
                print("  %-18s" % k, "  ".join("%s=%.4g" % (c.replace("SQ_", ""), sum(v[-last:]) / len(v[-last:])) for c, v in sorted(cs.items())))
if __name__ == "__main__":
    main(sys.argv[1:])
