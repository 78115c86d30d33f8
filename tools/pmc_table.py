"""Per-kernel totals of every counter in a rocprofv3 --pmc run (counter_collection CSVs under DIR), one row per kernel name.
usage: pmc_table.py DIR [name filter]"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(collections.Counter)
cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        k = re.sub(r"\(.*", "", k)
        k = re.sub(r"^void ", "", k).replace("gf::", "")
        if flt and flt not in k:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
names = sorted({c for v in tot.values() for c in v})
print("%-44s %8s " % ("kernel", "launches") + " ".join("%18s" % n for n in names))
for k, v in sorted(tot.items(), key=lambda kv: -max(kv[1].values())):
    print("%-44s %8d " % (k[:44], cnt[(k, names[0])]) + " ".join("%18.4g" % v[n] for n in names))
