"""A/B the cfg3 step under environment switches: python tools/ab.py "GF_X=1 GF_Y=2" "GF_X=0" ...  (each argument is one arm;
"" = defaults).  Prints ms per step and the per-kernel table of every arm."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for arm in sys.argv[1:] or [""]:
    env = dict(os.environ)
    for kv in arm.split():
        k, v = kv.split("=", 1)
        env[k] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-extra"],
                         capture_output=True, text=True, env=env)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if not line:
        print("ARM [%s] FAILED\n%s" % (arm, out.stderr[-2000:]))
        continue
    d = json.loads(line[0])
    k = d["roofline"].get("kernel_ms_per_step", {})
    print("ARM [%s]: %.3f ms/step | %s" % (arm, d["ms_per_step"], " ".join("%s=%.3f" % (a.replace("smpf_", ""), b) for a, b in list(k.items())[:14])), flush=True)
