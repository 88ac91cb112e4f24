#!/usr/bin/env python3
"""Which allocation does a GPU fault address belong to? Reads a SHODH_GUARD_LOG file (csrc/guard.hip: one line per allocation / free) and prints the
blocks whose reserved range (fences included) contains the address, or the nearest ones.
usage: tools/guard_lookup.py <alloc.log> <0xADDRESS>"""
import re
import sys


def main(path, addr):
    addr = int(addr, 16)
    pat = re.compile(r"^(A|F) (host|dev) user=(0x[0-9a-f]+) bytes=(\d+) mapped=\[(0x[0-9a-f]+),(0x[0-9a-f]+)\) reserved=\[(0x[0-9a-f]+),(0x[0-9a-f]+)\) at (\S+)")
    blocks, freed = {}, set()
    for n, ln in enumerate(open(path, errors="replace")):
        m = pat.match(ln)
        if not m:
            continue
        what, kind, user, nbytes, m0, m1, r0, r1, where = m.groups()
        user = int(user, 16)
        if what == "A":
            blocks[user] = dict(line=n, kind=kind, user=user, bytes=int(nbytes), m0=int(m0, 16), m1=int(m1, 16), r0=int(r0, 16), r1=int(r1, 16), where=where, freed=None)
        elif user in blocks:
            blocks[user]["freed"] = n
    hits = [b for b in blocks.values() if b["r0"] <= addr < b["r1"]]
    if not hits:
        hits = sorted(blocks.values(), key=lambda b: min(abs(addr - b["r0"]), abs(addr - b["r1"])))[:3]
        print("no reserved range contains 0x%x; nearest:" % addr)
    for b in hits:
        end = b["user"] + b["bytes"]
        rel = ("%d bytes past the end" % (addr - end)) if addr >= end else ("%d bytes before the start" % (b["user"] - addr)) if addr < b["user"] else "inside (offset %d)" % (addr - b["user"])
        print("%s block at %s: user=0x%x bytes=%d mapped=[0x%x,0x%x) %s; fault page is %s%s" % (
            b["kind"], b["where"], b["user"], b["bytes"], b["m0"], b["m1"], "FREED at log line %d" % b["freed"] if b["freed"] is not None else "live", rel,
            " (page granularity: the access is somewhere in the 4 KiB page starting there)"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
