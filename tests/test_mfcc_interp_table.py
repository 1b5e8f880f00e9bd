"""The tables of the MFCC bins interpolated inside the fused kernel (mfcc_interp_t, vbx_kernels.hpp), held to the frame's exact
DFT on the CPU: for a frame of n samples inside a transform of M >= 2 n, bin k of the n-point DFT (src/spectrum.rs:401-441 takes
|X_n[k]|) is the DTFT at k / n, and the kernel forms it from 32 of the transform's bins.  numpy plays the transform; the taps,
first-tap indices and rotations are the library's own host-built tables (no device call)."""
import ctypes as C

import numpy as np
import pytest


def _table(pkg, n, b_lo, nb):
    lib = pkg.load_library()
    fn = lib.vbx_internal_mfcc_interp_table
    fn.restype = C.c_int
    fn.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    desc = (C.c_int32 * 8)()
    need = C.c_size_t(0)
    rc = fn(n, b_lo, nb, desc, None, 0, C.byref(need))
    assert rc >= 0
    if rc == 0:
        return None
    buf = np.zeros(need.value, np.uint8)
    assert fn(n, b_lo, nb, desc, buf.ctypes.data, buf.size, C.byref(need)) == 1
    M, nt, slots, taps, jmin, jmax, o_coef, o_j0 = list(desc)
    rot = buf[:(M // 4 + 1) * 16].view(np.float64).reshape(-1, 2)
    coef = buf[o_coef:o_coef + slots * (taps // 2) * nt * 16].view(np.float64).reshape(slots, taps // 2, nt, 2)
    j0 = buf[o_j0:o_j0 + slots * nt * 4].view(np.int32).reshape(slots, nt)
    return dict(M=M, nt=nt, slots=slots, taps=taps, jmin=jmin, jmax=jmax, rot=rot[:, 0] + 1j * rot[:, 1], coef=coef, j0=j0)


def _interpolated_bins(t, x, nb):
    """what the kernel computes, in numpy: Z[j] = X_M[j] e^{2 pi i j c / M} (Z[-j] = conj Z[j]), bin = sum of taps * Z"""
    M = t["M"]
    Y = np.fft.fft(x, M)
    jj = np.arange(t["jmin"], t["jmax"] + 1)
    Z = np.where(jj >= 0, Y[np.abs(jj)] * t["rot"][np.abs(jj)], np.conj(Y[np.abs(jj)] * t["rot"][np.abs(jj)]))
    out = np.zeros(nb, complex)
    for b in range(nb):
        u, th = divmod(b, t["nt"])
        taps = t["coef"][u, :, th, :].reshape(-1)                      # taps 2 t, 2 t + 1 interleaved = tap order
        out[b] = np.dot(taps, Z[t["j0"][u, th]:t["j0"][u, th] + t["taps"]])
    return out


def _exact_bins(x, n, b_lo, nb):
    k = np.arange(b_lo, b_lo + nb, dtype=np.longdouble)[:, None]
    i = np.arange(n, dtype=np.longdouble)[None, :]
    ph = (k * i / n) % 1.0
    ang = 2.0 * np.pi * ph.astype(np.float64)                          # the phase reduced in long double, the sum in double
    return (np.cos(ang) - 1j * np.sin(ang)) @ x


SHAPES = [(1025, 2, 180), (1103, 2, 182), (1103, 0, 250), (1199, 3, 200), (700, 1, 117), (882, 2, 160), (1000, 2, 167), (1201, 2, 200),
          (1600, 3, 264), (2047, 4, 412), (2049, 4, 340), (3000, 6, 497), (4000, 8, 664), (4095, 8, 680)]


@pytest.mark.parametrize("n,b_lo,nb", SHAPES)
def test_interpolated_bins_are_the_frames_dft(pkg, n, b_lo, nb):
    t = _table(pkg, n, b_lo, nb)
    assert t is not None, "the shape should have the interpolated form"
    assert t["M"] >= 2 * n and t["M"] % n != 0 and t["taps"] in (24, 32, 40)
    rng = np.random.default_rng(n)
    i = np.arange(n)
    han = 0.5 - 0.5 * np.cos(2 * np.pi * i / (n - 1))
    signals = {
        "noise": rng.standard_normal(n),
        "tone between bins under a Hanning window (valleys 1e-10 of the peak)": np.cos(2 * np.pi * (b_lo + 20.37) / n * i + 0.3) * han,
        "harmonics + a little noise": sum(np.cos(2 * np.pi * 0.00417 * h * i) / h for h in range(1, 12)) * han + 1e-4 * rng.standard_normal(n),
        "impulses at both ends (the edges of the band limit)": np.where((i == 0) | (i == n - 1), 1.0, 0.0),
        "dc": np.ones(n),
    }
    for name, x in signals.items():
        xe = _exact_bins(x, n, b_lo, nb)
        xi = _interpolated_bins(t, x, nb)
        scale = np.abs(np.fft.fft(x, t["M"])).max()
        err = np.abs(np.abs(xi) - np.abs(xe)).max() / scale
        # the taps' design error (24 / 32 / 40 taps by M / n, Kaiser-Bessel bump) is held at the transform's own rounding: the worst frame,
        # an impulse at its very end, 4e-15
        assert err < 1e-14, (name, err)
        # the phase factor e^{i w c} that drops out of |X|^2 really is one: the complex values agree after it
        k = np.arange(b_lo, b_lo + nb, dtype=np.int64)
        ph = ((k * (n - 1)) % (2 * n)).astype(np.float64) / (2.0 * n)          # k c / n reduced exactly: c = (n - 1) / 2
        assert np.abs(xi - xe * np.exp(2j * np.pi * ph)).max() / scale < 5e-14, name


def test_shapes_without_the_form(pkg):
    # lengths that divide the transform take their bins from it directly; bins beyond a quarter of the transform have no taps
    assert _table(pkg, 1200, 2, 200) is None
    assert _table(pkg, 600, 1, 100) is None
    assert _table(pkg, 1024, 2, 170) is None
    assert _table(pkg, 1103, 300, 200) is None                         # up to bin 500 of 1103: beyond M / 4 = 600 of 2400
    # ... and every tap index stays inside [jmin, jmax], absent bins carry zero taps
    t = _table(pkg, 1103, 2, 182)
    assert t["j0"].min() >= 0 and (t["j0"] + t["taps"] - 1).max() <= t["jmax"] - t["jmin"]
    flat = t["coef"].transpose(0, 2, 1, 3).reshape(t["slots"] * t["nt"], -1)
    assert np.all(flat[182:] == 0.0) and np.all(np.abs(flat[:182]).sum(axis=1) > 0.5)
