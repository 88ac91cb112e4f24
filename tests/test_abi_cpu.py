"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/shodh_hip.h declares; host-side fusion math equals the oracle; and the product
fails loudly (no CPU fallback) when no GPU is present. No device compute here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shodh():
    from shodh_memory_amd import build
    build.build()
    import shodh_memory_amd
    return shodh_memory_amd


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "shodh_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(shodh_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_are_exported(shodh):
    from shodh_memory_amd import _lib
    names = declared_symbols()
    assert len(names) >= 40
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libshodh_hip.so does not export %s" % n
    # the python binding table covers exactly the header
    assert sorted(_lib.SYMBOLS) == names
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (shodh_[a-z0-9_]+)", out))
    assert set(names) <= exported


def test_abi_version_and_structs(shodh):
    from shodh_memory_amd import _lib
    assert _lib.lib().shodh_abi_version() == 5
    cfg = _lib.IndexCfg()
    _lib.lib().shodh_index_cfg_default(C.byref(cfg))
    assert (cfg.dim, cfg.metric, cfg.kind, cfg.order, cfg.nprobe) == (384, 0, 0, 0, 20)
    assert (cfg.max_degree, cfg.search_list_size, round(cfg.alpha, 6)) == (32, 75, 1.2)        # VamanaConfig::default, vamana.rs:79-90
    assert C.sizeof(_lib.IndexCfg) == 64 and C.sizeof(_lib.Weights) == 32


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "shodh_memory_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle under oracle/ is never used", "").replace(
                    "The oracle (oracle/) is never imported from here", "").lower() or f in ("__init__.py", "_lib.py"), \
                    "%s mentions the oracle" % f
                assert "import oracle" not in src and "from oracle" not in src


def test_no_device_fails_loudly(shodh):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from shodh_memory_amd import _lib
    with pytest.raises(_lib.ShodhError) as e:
        shodh.VamanaIndex(shodh.VamanaConfig(dimension=384))
    assert e.value.code == _lib.ERR_DEVICE and "no CPU fallback" in str(e.value)


def test_host_fusion_matches_oracle(shodh, oracle):
    w = shodh.LearnedWeights.default()
    ow = oracle.weights_default()
    assert w.as_tuple() == ow.as_tuple()
    rng = np.random.default_rng(1)
    for _ in range(500):
        s = rng.uniform(-0.2, 1.2, 4).astype(np.float32)
        mom = np.float32(rng.uniform(-1, 1)); acc = int(rng.integers(0, 40)); gs = np.float32(rng.uniform(0, 1))
        a = w.fuse_scores_full(*map(float, s), float(mom), acc, float(gs))
        b = oracle.fuse_scores_full(ow, *map(float, s), float(mom), acc, float(gs))
        assert abs(a - b) <= 1e-6
    assert abs(w.fuse_scores_full(.9, .9, .9, .9, .9, 16, .9) - 0.986757) < 1e-6
    assert w.fuse_scores(.9, .9, .9, .9) == w.fuse_scores_full(.9, .9, .9, .9, 0.0, 0, 0.5)
    assert w.fuse_scores_with_momentum(.9, .9, .9, .9, .3) == w.fuse_scores_full(.9, .9, .9, .9, .3, 0, 0.5)
    for bad in (float("nan"), float("inf"), float("-inf")):
        assert np.isfinite(w.fuse_scores_full(bad, .5, .5, .5, 0.0, 1, .5))
        assert shodh.calibrate_score(bad) == 0.0
    assert abs(shodh.calibrate_score(0.5) - 0.5) < 1e-3 and shodh.calibrate_score(0.9) > 0.9 and shodh.calibrate_score(0.1) < 0.1
    # feedback + normalize track the oracle exactly
    for args in [(False, True, False, False), (True, False, True, True), (False, False, False, True)]:
        w.apply_feedback(*args)
        oracle.weights_apply_feedback(ow, *args)
        assert w.as_tuple() == ow.as_tuple() and w.update_count == ow.update_count
    w2 = shodh.LearnedWeights(*([0.5] * 7))
    w2.normalize()
    assert all(abs(x - 1 / 7) < 1e-3 for x in w2.as_tuple())


def test_rust_bindings_are_current_and_complete():
    """rust/shodh-hip-sys/src/ffi.rs is generated from the header (tools/gen_rust_ffi.py): it must be up to date and
    declare every symbol the header declares."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "gen_rust_ffi.py"), "--check"]).returncode == 0
    ffi = open(os.path.join(root, "rust", "shodh-hip-sys", "src", "ffi.rs")).read()
    assert sorted(re.findall(r"pub fn (shodh_\w+)\(", ffi)) == declared_symbols()


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/shodh_hip.h must be consumable by a C compiler (the boundary is a C ABI: cgo / bindgen / ctypes users), and a
    C program must link against the library and call entry points that need no device."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "shodh_hip.h")
    assert subprocess.run(["gcc", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", hdr]).returncode == 0
    assert subprocess.run(["g++", "-x", "c++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", hdr]).returncode == 0
    src = tmp_path / "t.c"
    src.write_text('''
#include <stdio.h>
#include "shodh_hip.h"
int main(void) {
    shodh_weights w; shodh_weights_default(&w);
    float s = shodh_fuse_scores_full(&w, 0.9f, 0.9f, 0.9f, 0.9f, 0.9f, 16, 0.9f);
    float d[3]; shodh_density_weights(0.3f, d);
    const char *tags[2] = {"rust", "python"};
    float t = shodh_calculate_tag_score("Learning Rust", tags, 2);
    shodh_leg_fusion_cfg c; shodh_leg_fusion_cfg_default(&c);
    printf("%d %.6f %.3f %.2f %.1f\\n", shodh_abi_version(), s, d[1], t, c.rrf_k);
    return 0;
}
''')
    exe = tmp_path / "t"
    libdir = os.path.join(root, "shodh_memory_amd")
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), "-L", libdir, "-lshodh_hip",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ver, s, g, t, k = out.stdout.split()
    assert abs(float(s) - 0.986757) < 1e-5 and float(g) == 0.5 and float(t) == 0.5 and float(k) == 30.0 and int(ver) >= 1


def test_prebuilt_library_is_trusted_by_content_not_by_mtime(shodh, tmp_path, monkeypatch):
    """build() reuses a library only when the hash recorded next to it equals the hash of the sources + flags on disk (a snapshot pushed to another box
    keeps no meaningful modification times, and the .so that rides along must provably belong to the sources next to it)."""
    from shodh_memory_amd import build as B
    B.build()
    assert not B.needs_build() and open(B.STAMP).read().strip() == B.source_hash()
    real = B._deps
    extra = tmp_path / "extra.h"
    extra.write_text("// a changed source\n")
    monkeypatch.setattr(B, "_deps", lambda: real() + [str(extra)])
    assert B.needs_build()                                      # same mtimes, different content
    monkeypatch.setattr(B, "_deps", real)
    os.utime(B.LIB, (1, 1))                                     # an ancient library with the right content is still fine
    assert not B.needs_build()
