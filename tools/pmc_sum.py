"""Sum a rocprofv3 --pmc counter per kernel name from the counter_collection CSV(s) under a directory.
usage: pmc_sum.py DIR COUNTER [steps]   -> prints per-kernel total (KiB -> GB, raw) and per step"""
import csv, glob, sys, collections, re
d, ctr = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tot = collections.Counter(); cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != ctr:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        k = re.sub(r"\(.*", "", k)
        k = re.sub(r"^void ", "", k)
        k = re.sub(r"<.*", "", k)
        k = k.split("::")[-1]
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:30]:
    print(f"{k:40s} launches {cnt[k]:6d}  {ctr} {v*1024/1e9/steps:9.3f} GB/step (raw KiB x1)")
