"""rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv) -> markdown table for profiles/. usage: stats_to_md.py DIR_OR_CSV TITLE > out.md"""
import csv, glob, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
print("# %s\n" % (sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 --kernel-trace --stats"))
print("Names truncated to 100 chars; durations in ns. Raw CSV next to this file.\n")
print("| kernel | calls | total ns | avg ns | % | min ns | max ns |\n|---|---|---|---|---|---|---|")
for r in rows:
    print("| `%s` | %s | %s | %s | %s | %s | %s |" % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"].split(".")[0], r["Percentage"], r["MinNs"], r["MaxNs"]))
