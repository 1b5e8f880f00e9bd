"""ctypes binding of the CPU oracle (oracle/libvbx_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VBX_ORACLE_ASAN=1: the AddressSanitizer + UBSan build (tests/test_sanitizers.py; the process needs libasan preloaded)
_ASAN = os.environ.get("VBX_ORACLE_ASAN") == "1"
_SO = os.path.join(_HERE, "libvbx_oracle_asan.so" if _ASAN else "libvbx_oracle.so")

OK, ERR_LPC, ERR_POLYNOMIAL, ERR_NAN, ERR_PANIC, ERR_WORKSPACE = range(6)
MAX_RESONANCES = 32


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("vbx_oracle.c", "vbx_oracle_f32.c", "vbx_cpu_bench.c", "vbx_soak.c", "vbx_oracle.h")]
    if force or not os.path.exists(_SO) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in src
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if _ASAN else []))
    return _SO


_lib = None
_dp = C.POINTER(C.c_double)


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("autocorr_macs", "sinc_terms", "sinc_evals", "brent_calls", "candidates")]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        L = _lib
        L.vbxo_max_amplitude.restype = C.c_double
        L.vbxo_rms.restype = C.c_double
        L.vbxo_hz_to_mel.restype = C.c_double
        L.vbxo_mel_to_hz.restype = C.c_double
        L.vbxo_hz_to_mel.argtypes = [C.c_double]
        L.vbxo_mel_to_hz.argtypes = [C.c_double]
        L.vbxo_degree.restype = C.c_size_t
        L.vbxo_off_low.restype = C.c_size_t
        L.vbxo_to_resonance.restype = C.c_size_t
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def window(kind, n):
    w = np.empty(n, dtype=np.float64)
    fn = {"hanning": "vbxo_window_hanning", "hanning_lag": "vbxo_window_hanning_lag",
          "hanning_periodic": "vbxo_window_hanning_periodic"}[kind]
    getattr(lib(), fn)(_p(w), C.c_size_t(n))
    return w


def sine(n, rate, hz):
    x = np.empty(n, dtype=np.float64)
    lib().vbxo_sine(_p(x), C.c_size_t(n), C.c_double(rate), C.c_double(hz))
    return x


def max_amplitude(x):
    x = _f64(x)
    return lib().vbxo_max_amplitude(_p(x), C.c_size_t(x.size))


def normalize(x):
    x = _f64(x).copy()
    lib().vbxo_normalize(_p(x), C.c_size_t(x.size))
    return x


def rms(x):
    x = _f64(x)
    return lib().vbxo_rms(_p(x), C.c_size_t(x.size))


def preemphasis(x, factor):
    x = _f64(x).copy()
    lib().vbxo_preemphasis(_p(x), C.c_size_t(x.size), C.c_double(factor))
    return x


def autocorrelate(x, n_lags):
    x = _f64(x)
    out = np.empty(n_lags, dtype=np.float64)
    lib().vbxo_autocorrelate(_p(x), C.c_size_t(x.size), _p(out), C.c_size_t(n_lags))
    return out


def interpolate_sinc(y, offset, nx, x, depth):
    y = _f64(y)
    out = C.c_double()
    st = lib().vbxo_interpolate_sinc(_p(y), C.c_size_t(y.size), C.c_long(offset), C.c_size_t(nx),
                                     C.c_double(x), C.c_size_t(depth), C.byref(out))
    return st, out.value


def improve_extremum_sinc(y, offset, nx, ixmid, depth):
    y = _f64(y)
    xm, ym = C.c_double(), C.c_double()
    st = lib().vbxo_improve_extremum_sinc(_p(y), C.c_size_t(y.size), C.c_long(offset), C.c_size_t(nx),
                                          C.c_double(ixmid), C.c_size_t(depth), C.byref(xm), C.byref(ym))
    return st, xm.value, ym.value


def improve_extremum(y, offset, nx, ixmid, interp, depth=0, is_max=True):
    """improve_extremum with any Interpolation arm: interp 0 None, 1 Parabolic, 2 Sinc(depth).  Returns (status, xmid, ymid)."""
    y = _f64(y)
    xm, ym = C.c_double(), C.c_double()
    st = lib().vbxo_improve_extremum(_p(y), C.c_size_t(y.size), C.c_long(offset), C.c_size_t(nx), C.c_double(ixmid),
                                     C.c_int(interp), C.c_size_t(depth), C.c_int(1 if is_max else 0), C.byref(xm), C.byref(ym))
    return st, xm.value, ym.value


def pitch(x, sample_rate, threshold, fmin, fmax, cap=None):
    """Returns (status, candidates[count,2]) -- (frequency, strength), sorted as the reference."""
    x = _f64(x)
    cap = cap if cap is not None else x.size // 2 + 2
    out = np.zeros((cap, 2), dtype=np.float64)
    cnt = C.c_size_t()
    st = lib().vbxo_pitch(_p(x), C.c_size_t(x.size), C.c_double(sample_rate), C.c_double(threshold),
                          C.c_double(fmin), C.c_double(fmax), _p(out), C.c_size_t(cap), C.byref(cnt))
    return st, out[:min(cnt.value, cap)].copy(), cnt.value


def lpc(r, n_coeffs):
    r = _f64(r)
    assert r.size >= n_coeffs + 1
    out = np.empty(n_coeffs + 1, dtype=np.float64)
    lib().vbxo_lpc(_p(r), C.c_size_t(n_coeffs), _p(out))
    return out


def lpc_burg(x, n_coeffs):
    x = _f64(x)
    out = np.zeros(n_coeffs, dtype=np.float64)
    st = lib().vbxo_lpc_burg(_p(x), C.c_size_t(x.size), C.c_size_t(n_coeffs), _p(out))
    return st, out


def _c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def degree(p):
    p = _c128(p)
    return lib().vbxo_degree(_p(p), C.c_size_t(p.size))


def off_low(p):
    p = _c128(p)
    return lib().vbxo_off_low(_p(p), C.c_size_t(p.size))


class _C64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


def laguerre(p, start):
    p = _c128(p)
    L = lib()
    L.vbxo_laguerre.restype = _C64
    L.vbxo_laguerre.argtypes = [C.c_void_p, C.c_size_t, _C64]
    z = L.vbxo_laguerre(_p(p), p.size, _C64(start.real, start.imag))
    return complex(z.re, z.im)


def find_roots(p):
    p = _c128(p)
    roots = np.zeros(p.size, dtype=np.complex128)
    n = C.c_size_t()
    st = lib().vbxo_find_roots(_p(p), C.c_size_t(p.size), _p(roots), C.byref(n))
    return st, roots[:n.value].copy()


def find_roots_mut(p):
    p = _c128(p).copy()
    st = lib().vbxo_find_roots_mut(_p(p), C.c_size_t(p.size))
    return st, p


def div_polynomial(p, other):
    """div_polynomial_mut (src/polynomial.rs:155-195): returns (status, quotient-in-self, rem)."""
    p = _c128(p).copy()
    rem = np.zeros_like(p)
    L = lib()
    L.vbxo_div_polynomial_mut.argtypes = [C.c_void_p, C.c_size_t, _C64, C.c_void_p]
    st = L.vbxo_div_polynomial_mut(_p(p), p.size, _C64(other.real, other.imag), _p(rem))
    return st, p, rem


class _C32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


def laguerre_f32(p, start):
    """Complex<f32> instantiation (src/polynomial.rs:336-386)."""
    p = np.ascontiguousarray(p, dtype=np.complex64)
    L = lib()
    L.vbxo_laguerre_f32.restype = _C32
    L.vbxo_laguerre_f32.argtypes = [C.c_void_p, C.c_size_t, _C32]
    z = L.vbxo_laguerre_f32(_p(p), p.size, _C32(start.real, start.imag))
    return np.complex64(complex(z.re, z.im))


def find_roots_f32(p):
    p = np.ascontiguousarray(p, dtype=np.complex64)
    roots = np.zeros(p.size, dtype=np.complex64)
    n = C.c_size_t()
    st = lib().vbxo_find_roots_f32(_p(p), C.c_size_t(p.size), _p(roots), C.byref(n))
    return st, roots[:n.value].copy()


def find_roots_mut_f32(p):
    p = np.ascontiguousarray(p, dtype=np.complex64).copy()
    st = lib().vbxo_find_roots_mut_f32(_p(p), C.c_size_t(p.size))
    return st, p


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def autocorrelate_f32(x, n_lags):
    x = _f32(x)
    out = np.empty(n_lags, dtype=np.float32)
    lib().vbxo_autocorrelate_f32(_p(x), C.c_size_t(x.size), _p(out), C.c_size_t(n_lags))
    return out


def normalize_f32(x):
    x = _f32(x).copy()
    lib().vbxo_normalize_f32(_p(x), C.c_size_t(x.size))
    return x


def lpc_f32(r, n_coeffs):
    """Returns (ac[n_coeffs + 1], kc[n_coeffs]) in f32 arithmetic."""
    r = _f32(r)
    ac, kc = np.empty(n_coeffs + 1, dtype=np.float32), np.empty(n_coeffs, dtype=np.float32)
    lib().vbxo_lpc_f32(_p(r), C.c_size_t(n_coeffs), _p(ac), _p(kc))
    return ac, kc


def lpc_burg_f32(x, n_coeffs):
    x = _f32(x)
    out = np.zeros(n_coeffs, dtype=np.float32)
    st = lib().vbxo_lpc_burg_f32(_p(x), C.c_size_t(x.size), C.c_size_t(n_coeffs), _p(out))
    return st, out


def mfcc_f32(x, num_coeffs, lo, hi, sr):
    x = _f32(x)
    out = np.zeros(num_coeffs, dtype=np.float32)
    st = lib().vbxo_mfcc_f32(_p(x), C.c_size_t(x.size), C.c_size_t(num_coeffs), C.c_double(lo), C.c_double(hi),
                             C.c_double(sr), _p(out))
    return st, out


def pitch_f32(x, sample_rate, threshold, fmin, fmax, cap=None):
    """Pitched<f32, f32>::pitch: (status, candidates[count, 2] as float64 holders of f32 values, count)."""
    x = _f32(x)
    cap = cap if cap is not None else x.size // 2 + 2
    out = np.zeros((cap, 2), dtype=np.float64)
    cnt = C.c_size_t()
    st = lib().vbxo_pitch_f32(_p(x), C.c_size_t(x.size), C.c_float(sample_rate), C.c_float(threshold), C.c_float(fmin),
                              C.c_float(fmax), _p(out), C.c_size_t(cap), C.byref(cnt))
    return st, out[:min(cnt.value, cap)].copy(), cnt.value


def to_resonance(roots, sample_rate):
    roots = _c128(roots)
    out = np.zeros((max(roots.size, 1), 2), dtype=np.float64)
    n = lib().vbxo_to_resonance(_p(roots), C.c_size_t(roots.size), C.c_double(sample_rate), _p(out))
    return out[:n].copy()


def estimate_formants(est, res):
    """est: [k,2] (frequency, bandwidth) in/out copy; res: [m,2]."""
    est = _f64(est).copy()
    res = _f64(res)
    lib().vbxo_estimate_formants(_p(est), C.c_size_t(est.shape[0]), _p(res), C.c_size_t(res.shape[0]))
    return est


def find_formants(x, sample_rate, n_coeffs, formants):
    """Returns (status, formants_out[k,2], resonances[32,2], burg_coeffs[p])."""
    x = _f64(x)
    f = _f64(formants).copy()
    res = np.zeros((MAX_RESONANCES, 2), dtype=np.float64)
    co = np.zeros(n_coeffs, dtype=np.float64)
    st = lib().vbxo_find_formants(_p(x), C.c_size_t(x.size), C.c_double(sample_rate), C.c_size_t(n_coeffs),
                                  _p(f), C.c_size_t(f.shape[0]), _p(res), _p(co))
    return st, f, res, co


def resampled_len(n, ratio):
    L = lib()
    L.vbxo_resampled_len.restype = C.c_size_t
    return L.vbxo_resampled_len(C.c_size_t(n), C.c_double(ratio))


def resample_linear(x, ratio):
    x = _f64(x)
    out = np.zeros(resampled_len(x.size, ratio), dtype=np.float64)
    lib().vbxo_resample_linear(_p(x), C.c_size_t(x.size), C.c_double(ratio), _p(out))
    return out


def find_formants_ratio(x, sample_rate, ratio, n_coeffs, formants):
    x = _f64(x)
    f = _f64(formants).copy()
    st = lib().vbxo_find_formants_ratio(_p(x), C.c_size_t(x.size), C.c_double(sample_rate), C.c_double(ratio),
                                        C.c_size_t(n_coeffs), _p(f), C.c_size_t(f.shape[0]))
    return st, f


def hz_to_mel(hz):
    return lib().vbxo_hz_to_mel(hz)


def mel_to_hz(mel):
    return lib().vbxo_mel_to_hz(mel)


def dct(signal):
    s = _f64(signal)
    out = np.empty_like(s)
    lib().vbxo_dct(_p(s), C.c_size_t(s.size), _p(out))
    return out


def mfcc_bins(n, num_coeffs, lo, hi, sr):
    out = np.zeros(num_coeffs + 2, dtype=np.uint64)
    lib().vbxo_mfcc_bins(C.c_size_t(n), C.c_size_t(num_coeffs), C.c_double(lo), C.c_double(hi),
                         C.c_double(sr), _p(out))
    return out.astype(np.int64)


def mfcc(x, num_coeffs, lo, hi, sr, use_fft=False):
    x = _f64(x)
    out = np.zeros(num_coeffs, dtype=np.float64)
    st = lib().vbxo_mfcc(_p(x), C.c_size_t(x.size), C.c_size_t(num_coeffs), C.c_double(lo), C.c_double(hi),
                         C.c_double(sr), _p(out), C.c_int(1 if use_fft else 0))
    return st, out


def fft(x):
    x = _c128(x)
    out = np.empty_like(x)
    lib().vbxo_fft(_p(x), _p(out), C.c_size_t(x.size))
    return out


def counters_reset():
    lib().vbxo_counters_reset()


def counters():
    c = Counters()
    lib().vbxo_counters_get(C.byref(c))
    return {n: getattr(c, n) for n, _ in Counters._fields_}


# ---- batch helpers used by the parity tests (frame loops = the reference's L4 user code) ----

def frames_view(audio, n, hop):
    """Windower semantics: frame t = audio[t*hop : t*hop+n] while n <= remaining."""
    audio = _f64(audio)
    f = (audio.size - n) // hop + 1 if audio.size >= n else 0
    return np.lib.stride_tricks.as_strided(audio, shape=(f, n), strides=(hop * 8, 8), writeable=False)


def cpu_bench(workload, audio, frame_len, hop, order, sample_rate, n_threads, seconds, max_frames=0):
    """vbx_cpu_bench.c: the oracle's frame loop on native threads.  workload: 'pipeline' | 'config2' | 'config3' |
    'config4'.  Returns (frames_done, elapsed_seconds)."""
    a = _f64(audio)
    code = {"pipeline": 0, "config2": 1, "config3": 2, "config4": 3}[workload]
    done, dt, cs = C.c_ulong(), C.c_double(), C.c_double()
    fn = lib().vbxo_cpu_bench
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_int, C.c_double,
                   C.c_ulong, C.POINTER(C.c_ulong), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rc = fn(code, _p(a), a.size, frame_len, hop, order, sample_rate, n_threads, seconds, max_frames,
            C.byref(done), C.byref(dt), C.byref(cs))
    if rc != 0:
        raise ValueError("vbxo_cpu_bench: bad argument")
    return int(done.value), float(dt.value)


def usable_cores():
    """Threads worth starting: the affinity mask capped by the cgroup CPU quota (a container with 256 logical CPUs and a
    16-CPU quota runs 256 threads at 1/16 speed each)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


SOAK_PITCH, SOAK_LPC, SOAK_MFCC, SOAK_FORMANTS = 1, 2, 4, 8


def soak(audio, frame_len, hop, first, count, order, sample_rate, what, n_threads=None, pitch=(0.2, 75.0, 600.0),
         mfcc=(13, 100.0, 8000.0)):
    """vbx_soak.c: frames [first, first + count) of the hop-strided view through the oracle on native threads; every
    per-frame result of the parts selected by `what` (SOAK_* bits) as numpy arrays."""
    a = _f64(audio)
    n_threads = n_threads or usable_cores()
    p, k = order, mfcc[0]
    out = {}
    i32 = lambda *sh: np.zeros(sh, dtype=np.int32)
    f64 = lambda *sh: np.zeros(sh, dtype=np.float64)
    if what & SOAK_PITCH:
        out.update(pitch_status=i32(count), pitch_count=i32(count), pitch_top=f64(count, 3, 2))
    if what & SOAK_LPC:
        out.update(r=f64(count, p + 1), a=f64(count, p + 1))
    if what & SOAK_MFCC:
        out.update(mfcc=f64(count, k), mfcc_status=i32(count))
    if what & SOAK_FORMANTS:
        out.update(burg=f64(count, p), ff_status=i32(count), res=f64(count, MAX_RESONANCES, 2), res_count=i32(count))
    g = lambda name: _p(out[name]) if name in out else None
    fn = lib().vbxo_soak
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_int, C.c_int,
                   C.c_double, C.c_double, C.c_double, C.c_size_t, C.c_double, C.c_double] + [C.c_void_p] * 11
    rc = fn(_p(a), a.size, frame_len, hop, first, count, order, sample_rate, what, n_threads,
            pitch[0], pitch[1], pitch[2], k, mfcc[1], mfcc[2],
            g("pitch_status"), g("pitch_count"), g("pitch_top"), g("r"), g("a"), g("mfcc"), g("mfcc_status"),
            g("burg"), g("ff_status"), g("res"), g("res_count"))
    if rc != 0:
        raise ValueError("vbxo_soak: bad argument")
    return out


def soak_track(res, ff_status, est_init, seg_start=None):
    """vbx_soak.c: the sequential formant tracker over resonance rows [F, 32, 2] (restarts at seg_start entries)."""
    res = _f64(res)
    st = np.ascontiguousarray(ff_status, dtype=np.int32)
    est = _f64(est_init)
    seg = None if seg_start is None else np.ascontiguousarray(seg_start, dtype=np.int64)
    out = np.zeros((res.shape[0], est.shape[0], 2), dtype=np.float64)
    fn = lib().vbxo_soak_track
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    rc = fn(_p(res), st.ctypes.data, res.shape[0], None if seg is None else seg.ctypes.data, 0 if seg is None else seg.size,
            _p(est), est.shape[0], _p(out))
    if rc != 0:
        raise ValueError("vbxo_soak_track: bad argument")
    return out
