"""Average several rocprofv3 --pmc counters per kernel: pmc_avg.py DIR [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(lambda: collections.Counter()); cnt = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if sub not in k: continue
        k = k.split("(")[0][-60:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]): print(f"   {c:32s} {tot[k][c]/cnt[k][c]:16.0f}  (x{cnt[k][c]})")
