"""Per-dispatch durations of one training step from a rocprofv3 --kernel-trace CSV: trace_step.py DIR [name-substring]"""
import csv, glob, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", ""), r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
rows.sort()
# last step = dispatches after the last zero_f32
last = max(i for i, r in enumerate(rows) if "zero_f32" in r[2])
# find the forward start: previous readout_molecules..., simply take the window between the previous zero_f32 and last
prev = max(i for i, r in enumerate(rows[:last]) if "zero_f32" in r[2])
for s, e, n, g in rows[prev:last]:
    if sub in n:
        print(f"{(e-s)/1e3:9.1f} us  grid {g:>10s}  {n.split('(')[0][-70:]}")
