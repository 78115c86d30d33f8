"""end_to_end / device-step lines of bench.py JSON outputs: python tools/e2e_ab.py a.json b.json ..."""
import json
import sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    e = d.get("end_to_end", {})
    print("%-28s step %.3f ms   end_to_end %.3f ms   ratio %.3f" % (f.split("/")[-1], d["ms_per_step"], e.get("ms_per_step", 0), e.get("ms_per_step", 0) / d["ms_per_step"]))
