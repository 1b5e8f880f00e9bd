"""SURVEY section 5 (sanitizers): the CPU side under AddressSanitizer + UndefinedBehaviorSanitizer.
* the oracle (oracle/libvbx_oracle_asan.so, `make -C oracle asan`): its known-answer tests and the soak walker's tests re-run
  on the instrumented build;
* the library's host-only entry points (csrc/vbx_host.cpp -> lib/libvbx_host_asan.so, `make -C vox_box.rs_amd host_asan`):
  tools/host_property_check.py -- random worlds / rows / segment lists / table sizes / mel geometries and misuse.
The GPU kernels cannot run under a sanitizer on this pool (no GPU ASan / XNACK); these are the parts that can."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_env():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("gcc ships no libasan.so here")
    env = dict(os.environ)
    env.update({"LD_PRELOAD": asan, "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    return env


def _clean(out):
    assert "AddressSanitizer" not in out and "runtime error:" not in out and "LeakSanitizer" not in out, out[-4000:]


def test_oracle_known_answer_and_soak_tests_under_asan_ubsan():
    env = _asan_env()
    env["VBX_ORACLE_ASAN"] = "1"
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_kat.py"), os.path.join(ROOT, "tests", "test_oracle_soak.py")],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    out = r.stdout + r.stderr
    _clean(out)
    assert r.returncode == 0, out[-4000:]
    assert " passed" in out


def test_oracle_asan_build_is_the_one_loaded():
    """The switch works: with VBX_ORACLE_ASAN=1 pyoracle loads the instrumented library (not the -O2 one)."""
    env = _asan_env()
    env["VBX_ORACLE_ASAN"] = "1"
    code = ("import sys; sys.path.insert(0, %r); import pyoracle as o; o.build(); o.lib(); "
            "maps = open('/proc/self/maps').read(); print('ASAN_LIB' if 'libvbx_oracle_asan.so' in maps else 'PLAIN_LIB')") % os.path.join(ROOT, "oracle")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ASAN_LIB" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("seed", [1, 2])
def test_host_entry_points_under_asan_ubsan(seed):
    env = _asan_env()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "vox_box.rs_amd"), "-s", "host_asan"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_property_check.py"), str(seed)], env=env, capture_output=True,
                       text=True, timeout=900)
    out = r.stdout + r.stderr
    _clean(out)
    assert r.returncode == 0 and "host property test: ok" in out, out[-4000:]


def test_mfcc_bins_match_the_oracle(pkg, oracle):
    """vbx_mfcc_bins (host) == the oracle's restatement of src/spectrum.rs:411-414, incl. the 48 kHz bins SURVEY 8a lists."""
    import numpy as np
    b, bad = pkg.mfcc_bins(1200, 13, 100.0, 8000.0, 48000.0)
    assert list(b) == [2, 6, 11, 17, 24, 32, 42, 54, 69, 86, 107, 133, 163, 200, 244] and not bad
    for n, k, lo, hi, sr in ((512, 13, 100.0, 8000.0, 48000.0), (1024, 26, 133.0, 6855.0, 22050.0), (2048, 13, 100.0, 4000.0, 11025.0),
                             (1103, 20, 0.0, 8000.0, 44100.0), (31232, 13, 100.0, 4000.0, 11025.0)):
        b, bad = pkg.mfcc_bins(n, k, lo, hi, sr)
        assert np.array_equal(b, oracle.mfcc_bins(n, k, lo, hi, sr)) and not bad
    assert pkg.mfcc_bins(64, 13, 100.0, 30000.0, 22050.0)[1]           # bins beyond the spectrum: the reference panics
