"""Timeline of ONE search step from a rocprofv3 --kernel-trace CSV: every kernel between two consecutive launches of `marker`
(default: the last kernel of a step), with its start offset, duration and the idle gap before it.
    python tools/step_timeline.py <dir with *kernel_trace.csv> [marker substring] [which occurrence]"""
import csv, glob, sys
d = sys.argv[1]; marker = sys.argv[2] if len(sys.argv) > 2 else "lm_merge_kernel"; which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[which - 1], idx[which]
t0 = int(rows[a]["End_Timestamp"]); prev = t0
print(f"step = {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us, {b - a} kernels")
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:90]}")
    prev = e
