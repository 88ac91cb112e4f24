"""Build profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).
usage: pmc_traffic.py FETCH_DIR WRITE_DIR ROWS DIM NQ K > profiles/pmc_traffic.json
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of a wide streaming read -> doubled; units KB."""
import collections, csv, glob, json, os, sys
def load(d, counter):
    path = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "shodh" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-latency`",
       "correction": "gfx950: FETCH_SIZE reports exactly half of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section) -> doubled; units KB; WRITE_SIZE taken as is",
       "config": {"rows": int(sys.argv[3]), "dim": int(sys.argv[4]), "nq": int(sys.argv[5]), "k": int(sys.argv[6])},
       # the counters belong to THIS version of the kernel's source: bench.py refuses to quote them for another one
       "scan_mfma_hip_sha256": __import__("hashlib").sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shodh_memory_amd", "csrc", "scan_mfma.hip"), "rb").read()).hexdigest(),
       "kernels": {}}
for k in sorted(set(f) | set(w)):
    fr, wr = f.get(k, 0.0), w.get(k, 0.0)
    out["kernels"][k] = {"fetch_size_kb_raw": round(fr, 1), "write_size_kb_raw": round(wr, 1), "traffic_bytes_per_launch": int(fr * 2 * 1024 + wr * 1024)}
print(json.dumps(out, indent=1))
