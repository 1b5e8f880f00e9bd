#!/usr/bin/env python3
"""Property test of the library's host-only entry points (csrc/vbx_host.cpp) on its AddressSanitizer + UBSan build
(vox_box.rs_amd/lib/libvbx_host_asan.so; run with libasan preloaded -- tests/test_sanitizers.py does): random worlds, row
counts, segment lists, window sizes and mel geometries, with the invariants each function documents.  Any sanitizer report
aborts the process; the last line printed is "host property test: ok <n> cases"."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "vox_box.rs_amd", "lib", "libvbx_host_asan.so"))
sz, i32, dbl, vp = C.c_size_t, C.c_int, C.c_double, C.c_void_p


class Plan(C.Structure):
    _fields_ = [("lo", sz), ("hi", sz), ("warm", sz), ("stop", sz), ("continues_prev", i32), ("continues_next", i32)]


L.vbx_gather_plan.argtypes = [vp, i32, i32, i32, sz, vp, vp, vp]
L.vbx_shard_range.argtypes = [sz, i32, i32, vp, sz, C.POINTER(sz), C.POINTER(sz)]
L.vbx_shard_samples.argtypes = [sz, sz, sz, sz, C.POINTER(sz), C.POINTER(sz)]
L.vbx_shard_plan.argtypes = [sz, i32, i32, vp, sz, C.POINTER(Plan)]
L.vbx_shard_local_segments.argtypes = [C.POINTER(Plan), vp, sz, vp, sz, C.POINTER(sz)]
L.vbx_window_table_f64.argtypes = [i32, sz, vp]
L.vbx_window_table_f32.argtypes = [i32, sz, vp]
L.vbx_mfcc_bins.argtypes = [sz, sz, dbl, dbl, dbl, vp]
L.vbx_frame_count.argtypes = [sz, sz, sz]; L.vbx_frame_count.restype = sz
L.vbx_resampled_len.argtypes = [sz, dbl]; L.vbx_resampled_len.restype = sz
L.vbx_degree_c64.argtypes = [vp, sz]; L.vbx_degree_c64.restype = sz
L.vbx_off_low_c64.argtypes = [vp, sz]; L.vbx_off_low_c64.restype = sz
L.vbx_hz_to_mel.argtypes = [dbl]; L.vbx_hz_to_mel.restype = dbl
L.vbx_mel_to_hz.argtypes = [dbl]; L.vbx_mel_to_hz.restype = dbl

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1234)
cases = 0

for _ in range(400):                                   # gather plan: offsets tile the output, ops pair up
    world = int(rng.integers(1, 17))
    rows = rng.integers(0, 5000, world).astype(np.int64)
    if rng.random() < 0.3:
        rows[rng.integers(0, world)] = 0
    rec = int(rng.integers(1, 80))
    dst = int(rng.integers(0, world))
    plans = []
    for r in range(world):
        off, cnt, op = np.zeros(world, np.int64), np.zeros(world, np.int64), np.zeros(world, np.int32)
        assert L.vbx_gather_plan(rows.ctypes.data, world, r, dst, rec, off.ctypes.data, cnt.ctypes.data, op.ctypes.data) == 0
        plans.append((off, cnt, op))
    off, cnt, op = plans[dst]
    assert off[0] == 0 and np.array_equal(cnt, rows * rec) and np.array_equal(off[1:], np.cumsum(cnt)[:-1])
    for r in range(world):
        if r != dst:
            assert (plans[r][2][dst] == 2) == (rows[r] > 0) and (op[r] == 1) == (rows[r] > 0)
    assert L.vbx_gather_plan(rows.ctypes.data, world, world, dst, rec, None, None, None) < 0      # misuse is an error code
    assert L.vbx_gather_plan(None, world, 0, dst, rec, None, None, None) < 0
    cases += 1

for _ in range(600):                                   # shard geometry
    n = int(rng.integers(0, 200000)) if rng.random() < 0.9 else int(rng.integers(0, 40))
    world = int(rng.integers(1, 17))
    kind = rng.random()
    if kind < 0.25 or n == 0:
        seg = None
    else:
        k = int(rng.integers(1, 40))
        seg = np.unique(np.concatenate([[0], rng.integers(0, max(n, 1), k)])).astype(np.int64)
    sp, sn = (None, 0) if seg is None else (seg.ctypes.data, seg.size)
    prev_hi, prev_next = 0, 0
    for r in range(world):
        lo, hi = sz(), sz()
        assert L.vbx_shard_range(n, world, r, sp, sn, C.byref(lo), C.byref(hi)) == 0
        assert lo.value == prev_hi and hi.value >= lo.value
        pl = Plan()
        assert L.vbx_shard_plan(n, world, r, sp, sn, C.byref(pl)) == 0
        assert (pl.lo, pl.hi) == (lo.value, hi.value)
        if pl.hi > pl.lo:
            assert pl.warm <= 64 and pl.warm <= pl.lo and pl.warm <= pl.stop <= pl.hi - pl.lo + pl.warm
            assert pl.continues_prev == prev_next
            if pl.continues_prev:
                assert pl.warm == 64
            s0, s1 = sz(), sz()
            assert L.vbx_shard_samples(pl.lo - pl.warm, pl.hi, 1200, 480, C.byref(s0), C.byref(s1)) == 0
            assert s1.value - s0.value == (pl.hi - pl.lo + pl.warm - 1) * 480 + 1200
            need = sz()
            assert L.vbx_shard_local_segments(C.byref(pl), sp, sn, None, 0, C.byref(need)) == 0
            out = np.zeros(need.value, np.int64)
            assert L.vbx_shard_local_segments(C.byref(pl), sp, sn, out.ctypes.data, out.size, C.byref(need)) == 0
            assert out[0] == 0 and np.all(np.diff(out) > 0) and (out.size == 1 or out[-1] < pl.hi - pl.lo + pl.warm)
            if out.size > 1:                            # a too-small buffer is refused, not overrun
                small = np.zeros(out.size - 1, np.int64)
                assert L.vbx_shard_local_segments(C.byref(pl), sp, sn, small.ctypes.data, small.size, C.byref(need)) < 0
            prev_next = pl.continues_next
        else:
            assert pl.continues_prev == 0 and pl.continues_next == 0
        prev_hi = hi.value
    assert prev_hi == n and prev_next == 0
    cases += 1

for _ in range(300):                                   # tables
    n = int(rng.integers(1, 5000))
    for kind in range(4):
        t = np.full(n + 2, 7.0)
        assert L.vbx_window_table_f64(kind, n, t[1:].ctypes.data) == 0
        assert t[0] == 7.0 and t[-1] == 7.0 and (n < 3 or np.all(np.isfinite(t[1:-1])) or kind == 1)
        t32 = np.full(n + 2, 7.0, np.float32)
        assert L.vbx_window_table_f32(kind, n, t32[1:].ctypes.data) == 0 and t32[0] == 7.0 and t32[-1] == 7.0
    assert L.vbx_window_table_f64(9, n, t.ctypes.data) < 0 and L.vbx_window_table_f64(0, 0, t.ctypes.data) < 0
    k = int(rng.integers(1, 65))
    lo, hi, sr = float(rng.uniform(0, 500)), float(rng.uniform(600, 30000)), float(rng.choice([8000.0, 11025.0, 16000.0, 44100.0, 48000.0]))
    b = np.full(k + 4, -5, np.int32)
    rc = L.vbx_mfcc_bins(n, k, lo, hi, sr, b[1:].ctypes.data)
    assert rc in (0, 1) and b[0] == -5 and b[k + 3] == -5
    if rc == 0:
        assert np.all(np.diff(b[1:k + 3]) >= 0) and b[k + 2] <= n
    assert L.vbx_frame_count(n, n + 1, 3) == 0 and L.vbx_frame_count(10 * n, n, n) == 10
    assert L.vbx_resampled_len(n, 1.0) == n
    poly = (rng.standard_normal(2 * 9) * (rng.random(2 * 9) < 0.5)).view(np.complex128)
    d, o = L.vbx_degree_c64(poly.ctypes.data, poly.size), L.vbx_off_low_c64(poly.ctypes.data, poly.size)
    nz = np.nonzero(poly)[0]
    assert (d, o) == ((int(nz[-1]), int(nz[0])) if nz.size else (0, 0))
    assert abs(L.vbx_mel_to_hz(L.vbx_hz_to_mel(hi)) - hi) < 1e-9 * hi
    cases += 1
print("host property test: ok", cases, "cases")
