"""Builds libshodh_hip.so (HIP, gfx950 only) in-tree with hipcc. No torch, no cmake.

    python -m shodh_memory_amd.build          # incremental
    python -m shodh_memory_amd.build --force
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libshodh_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the exact-order kernels must not fuse a*b+c (rustc never does)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"] + os.environ.get("SHODH_EXTRA_FLAGS", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


STAMP = LIB + ".srchash"      # sha256 of every source the library was built from + the flags (travels with the .so; git-ignored like it)


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc"))) + [os.path.join(HERE, "..", "include", "shodh_hip.h")]


def source_hash():
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in _deps():
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


def needs_build():
    """True unless the library on disk was built from exactly these sources with these flags. Decided by CONTENT (a hash written next to the library
    at build time), not by modification times: a snapshot copied to another machine keeps neither a meaningful mtime order nor a guarantee that the
    binary riding along belongs to the sources next to it (the driver boxes ran whatever .so the push carried)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    try:
        return open(STAMP).read().strip() != source_hash()
    except OSError:
        return True


LAST_BUILD_MODE = "not run"


def build(force=False, verbose=False):
    global LAST_BUILD_MODE
    if not force and not needs_build():
        LAST_BUILD_MODE = "up to date (library matches the source hash %s...)" % source_hash()[:12]
        return LIB
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and all(os.path.getmtime(obj) > os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h")))
                and os.path.getmtime(obj) > os.path.getmtime(os.path.join(HERE, "..", "include", "shodh_hip.h"))):
            continue
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, out.decode()))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB, "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    LAST_BUILD_MODE = "rebuilt (%d of %d objects recompiled)" % (len(procs), len(objs))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
