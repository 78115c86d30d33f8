"""MFMA-pipe utilisation per kernel from one rocprofv3 counter pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d DIR -- python bench.py --steps 2 --warmup 1 ...
    python tools/mfma_busy.py DIR > profiles/rNN_cfg3_mfma_busy.txt
One line per kernel that issues MFMAs: total device time, CU-busy cycles / (256 CUs x duration) = the clock the part sustained while
busy, and MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES).  bench.py reads the last column (roofline.mfma_util.busy_pmc)."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
tot = collections.defaultdict(collections.Counter)
dur = collections.Counter()


def short(k):
    k = k.replace("(anonymous namespace)::", "")
    k = re.sub(r"^void ", "", k).replace("gf::", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^>(]*>)?", k)
    name, targs = m.group(1), (m.group(2) or "")
    if name.startswith("smp_rowpanel"):
        return name + targs.split(",")[0] + ">" if targs else name   # direction only
    return name + ("<" if targs and name.startswith("gemm") else "")


for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES" and "End_Timestamp" in r:
            dur[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
if not dur:
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra")
print("# kernel  total_ms  CU-busy cycles / (256 CUs x duration) = sustained clock while busy [GHz]   MFMA busy = MFMA_BUSY / (4 SIMDs x CU-busy)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    mf, cu = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("SQ_BUSY_CU_CYCLES", 0.0)
    if mf <= 0 or cu <= 0 or dur[k] <= 0:
        continue
    print("%-30s %8.3f ms   %.2f GHz   MFMA busy %.2f" % (k, dur[k], cu / (256 * dur[k] * 1e-3) / 1e9, mf / (4 * cu)))
