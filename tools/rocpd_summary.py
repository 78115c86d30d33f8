#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2) rocpd sqlite database into the text summary we commit under profiles/.
usage: tools/rocpd_summary.py gpurun_out/<dir>/<name>_results.db > profiles/<name>.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main(path):
    c = sqlite3.connect(path)
    rows = list(c.execute("select * from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats : %s" % path)
    print("# %-108s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in rows:
        print("%-110s %8d %14.1f %12.2f %7.2f" % (short(name), calls, total, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1])
