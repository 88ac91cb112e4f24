"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel, mean of each counter per dispatch."""
import collections, csv, glob, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if "shodh" not in n:
        continue
    acc[n.replace("void ", "").replace("shodh::", "")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())), "n=%d" % len(next(iter(d.values()))))
