#!/bin/bash
cd $(dirname $0)/..
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
SHODH_BENCH_EXTRAS_INPROC=1 SHODH_HIP_LIB=$PWD/shodh_memory_amd/libshodh_hip.so.prof timeout 600 python bench.py --steps 2 --warmup 1 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq 2>/dev/null | grep "^final q" | tail -400 > /tmp/fin.txt
python3 - <<'P'
import re,collections
rows=[l for l in open('/tmp/fin.txt')]
pat=re.compile(r"final q (\d+) n (\d+) nf (\d+) \| load-q (\d+) selectA (\d+) window (\d+) rescore (\d+) selectB (\d+)")
acc=[]
for l in rows:
    m=pat.search(l)
    if m: acc.append([int(x) for x in m.groups()])
acc=[a for a in acc if a[1] < 2000]
import statistics as st
if acc:
    print("n =", len(acc), "median n", st.median(a[1] for a in acc), "nf", st.median(a[2] for a in acc))
    for i,name in enumerate(("load-q","selectA","window","rescore","selectB")):
        print(name, "median cycles", st.median(a[3+i] for a in acc))
P
