"""The coalescing front (shodh_memory_amd/csrc/combiner.h) on the host alone: tests/cpp/combiner_test.cpp is built with g++ and run with a fake
device pass. What the front must guarantee whatever the timing: every caller gets ITS OWN result, a lone caller is never batched or delayed,
a failing pass reaches every member with its message, no pass exceeds the unit limit. (How WELL it batches depends on the machine's wake-up
latency; the GPU numbers are in bench.py's concurrent_callers entry.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("combiner") / "combiner_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "shodh_memory_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "combiner_test.cpp"), "-o", out])
    return out


@pytest.mark.parametrize("threads,calls,pass_us,max_units", [(1, 100, 100, 256), (2, 200, 150, 256), (8, 200, 200, 256), (32, 100, 200, 256), (32, 100, 200, 8)])
def test_combiner_contract(exe, threads, calls, pass_us, max_units):
    r = subprocess.run([exe, str(threads), str(calls), str(pass_us), str(max_units)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_callers_share_passes(exe):
    """16 closed-loop callers against a 300 us pass: far fewer passes than calls (each pass serves several callers)"""
    r = subprocess.run([exe, "16", "200", "300", "256"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    mean = float([l for l in r.stdout.splitlines() if l.startswith("mean_members")][0].split()[1])
    assert mean >= 3.0, r.stdout
