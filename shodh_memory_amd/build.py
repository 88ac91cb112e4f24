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


def _object_key(src):
    """What an object file was compiled from: the flags, its source and every header / .inc it can include -- by CONTENT. (Reuse used to be decided by
    modification times, which ignored SHODH_EXTRA_FLAGS and the .inc files: after a diagnostic build, or with objects carried along in a copied
    snapshot, the next plain build linked stale objects and stamped the library as matching the production sources.)"""
    import hashlib
    h = hashlib.sha256(" ".join([HIPCC] + FLAGS).encode())
    for d in [src] + _deps()[len(sources()):]:
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """force (or SHODH_FORCE_BUILD=1 in the environment): compile every object from source and relink, whatever is on disk."""
    global LAST_BUILD_MODE
    force = force or os.environ.get("SHODH_FORCE_BUILD", "0") not in ("", "0")
    objdir = os.path.join(HERE, "build")
    # A fresh GPU box gets the linked library with the push but no object files (.gpurunignore): the first build() there compiles everything from
    # source once (~30 s), so that every box -- the driver's included -- proves that THESE sources build and runs the binary they produce, not
    # whatever the push carried ("no driver box has ever compiled the library from source"). SHODH_TRUST_PREBUILT=1 skips that
    # (the content hash still has to match).
    fresh_gpu_box = (os.path.exists("/dev/kfd") and not glob.glob(os.path.join(objdir, "*.o.key"))
                     and os.environ.get("SHODH_TRUST_PREBUILT", "0") in ("", "0"))
    if fresh_gpu_box and not needs_build():
        force = True
    if not force and not needs_build():
        LAST_BUILD_MODE = "up to date (library matches the source hash %s...)" % source_hash()[:12]
        return LIB
    objs = []
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        keyfile = obj + ".key"
        key = _object_key(src)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(keyfile) and open(keyfile).read().strip() == key:
            continue
        if os.path.exists(keyfile):
            os.remove(keyfile)              # (written again only after a successful compile)
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, keyfile, key, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for src, keyfile, key, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append("hipcc failed for %s:\n%s" % (src, out.decode()))
            continue
        with open(keyfile, "w") as f:
            f.write(key + "\n")
    tmp_lib = LIB + ".linking"
    try:
        if failed:
            raise RuntimeError("\n".join(failed))
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp_lib, "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp_lib, LIB)            # (the library on disk is replaced only by a complete one)
    except Exception as ex:
        if os.path.exists(tmp_lib):
            os.remove(tmp_lib)
        if fresh_gpu_box and not needs_build():
            # the once-per-box from-source compile did not go through (no room, no compiler ...): the library the push carried still matches these sources by
            # content, so it is used -- loudly -- rather than failing every test and bench on a box problem
            LAST_BUILD_MODE = "prebuilt library used (content hash matches) AFTER THE FROM-SOURCE BUILD FAILED on this box: %s" % str(ex)[:300]
            return LIB
        raise
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    LAST_BUILD_MODE = "%s from source: %d of %d objects compiled, library relinked (source hash %s...)" % (
        ("fresh GPU box: compiled" if fresh_gpu_box else "FORCED rebuild") if force else "rebuilt", len(procs), len(objs), source_hash()[:12])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
