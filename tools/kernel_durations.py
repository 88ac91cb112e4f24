"""Per-step kernel durations (us) from a rocprofv3 --kernel-trace CSV: a step ends with a launch of `marker` whose Grid_Size_X equals `grid` (threads);
one line per step, one column per kernel between two markers.
    python tools/kernel_durations.py <dir> <marker> <grid threads> [max steps]"""
import csv, glob, sys
d, marker, grid = sys.argv[1], sys.argv[2], int(sys.argv[3]); mx = int(sys.argv[4]) if len(sys.argv) > 4 else 30
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], []
for r in rows:
    cur.append(r)
    if marker in r["Kernel_Name"]:
        if int(r["Grid_Size_X"]) == grid: steps.append(cur)
        cur = []
short = lambda n: n.split("(")[0].split("::")[-1][:22]
for st in steps[-mx:]:
    names = [short(r["Kernel_Name"]) for r in st]
    print(" | ".join("%s %.1f" % (nm, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for nm, r in zip(names, st))[-600:])
