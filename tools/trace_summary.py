"""Summarise a rocprofv3 kernel_trace.csv: per kernel and grid size, count / min / median / max in us."""
import collections, csv, sys
import glob, os
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "shodh" in n:
        by[(n.replace("void ", "").replace("shodh::", "")[:44], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(by.items()):
    v.sort()
    print("%-46s grid %8s x %-4s n=%-4d min %7.1f  med %7.1f  max %7.1f us" % (k[0], k[1], k[2], len(v), v[0], v[len(v) // 2], v[-1]))
