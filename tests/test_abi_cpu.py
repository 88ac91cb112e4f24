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
    assert _lib.lib().shodh_abi_version() == 1
    cfg = _lib.IndexCfg()
    _lib.lib().shodh_index_cfg_default(C.byref(cfg))
    assert (cfg.dim, cfg.metric, cfg.kind, cfg.order, cfg.nprobe) == (384, 0, 0, 0, 20)
    assert C.sizeof(_lib.IndexCfg) == 48 and C.sizeof(_lib.Weights) == 32


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
