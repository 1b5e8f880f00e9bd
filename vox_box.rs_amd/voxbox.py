"""ctypes binding of libvoxbox_hip.so (include/voxbox_hip.h).

Method names follow the reference crate's traits (Autocorrelate::autocorrelate, LPC::lpc /
lpc_praat, Pitched::pitch, Polynomial::find_roots, ToResonance::to_resonance,
EstimateFormants / FormantExtractor, MFCC::mfcc, vox_box::find_formants); each takes a whole
batch of frames instead of one slice.  Arguments named ``x`` may be a host numpy array
(uploaded for the call) or a DeviceArray / raw device pointer (used in place).
"""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VBX_LIB_PATH: experiment hook (tools/experiments/ab.py times variant builds of the library side by side)
LIB_PATH = os.environ.get("VBX_LIB_PATH") or os.path.join(_HERE, "lib", "libvoxbox_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "voxbox_hip.h")

WINDOW_HANNING, WINDOW_HANNING_LAG, WINDOW_HANNING_PERIODIC, WINDOW_RECTANGLE = 0, 1, 2, 3
FRAME_OK, FRAME_ERR_LPC, FRAME_ERR_POLYNOMIAL, FRAME_ERR_NAN, FRAME_ERR_PANIC = 0, 1, 2, 3, 4
MALE_FORMANT_ESTIMATES = (320.0, 1440.0, 2760.0, 3200.0)      # src/lib.rs:27
FEMALE_FORMANT_ESTIMATES = (480.0, 1760.0, 3200.0, 3520.0)    # src/lib.rs:28
MAX_RESONANCES = 32
MAX_PITCH_CANDIDATES = 1026           # VBX_MAX_PITCH_CANDIDATES


def pitch_max_candidates(frame_len):
    """VBX_PITCH_MAX_CANDIDATES(frame_len): no frame's candidate Vec (src/periodic.rs:452-454) is longer."""
    return frame_len // 4 + 2


class VoxBoxError(RuntimeError):
    pass


class _Complex32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class _Complex(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class _Resonance(C.Structure):
    _fields_ = [("frequency", C.c_double), ("bandwidth", C.c_double)]


class ShardPlan(C.Structure):
    """vbx_shard_plan_t (include/voxbox_hip.h): one rank's part of a recording sharded by frame ranges."""
    _fields_ = [("lo", C.c_size_t), ("hi", C.c_size_t), ("warm", C.c_size_t), ("stop", C.c_size_t),
                ("continues_prev", C.c_int), ("continues_next", C.c_int)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class AnalysisParams(C.Structure):
    """vbx_analysis_params (include/voxbox_hip.h): the parts of the user's frame loop that vbx_analyze_frames_f64 runs."""
    _fields_ = [("sample_rate", C.c_double),
                ("pitch_threshold", C.c_double), ("pitch_fmin", C.c_double), ("pitch_fmax", C.c_double),
                ("lpc_order", C.c_size_t), ("formant_order", C.c_size_t), ("n_est", C.c_size_t),
                ("est_init", _Resonance * 6),
                ("mfcc_coeffs", C.c_size_t), ("mfcc_lo_hz", C.c_double), ("mfcc_hi_hz", C.c_double)]

    @classmethod
    def make(cls, sample_rate, pitch=(0.2, 75.0, 600.0), lpc_order=12, formant_order=12, est_init=None,
             mfcc=(13, 100.0, 8000.0)):
        p = cls()
        p.sample_rate = sample_rate
        p.pitch_threshold, p.pitch_fmin, p.pitch_fmax = pitch
        p.lpc_order = lpc_order
        p.formant_order = formant_order
        est = np.asarray(est_init if est_init is not None else [[f, 1.0] for f in MALE_FORMANT_ESTIMATES], dtype=np.float64)
        p.n_est = est.shape[0] if formant_order else 0
        for i in range(int(p.n_est)):
            p.est_init[i].frequency, p.est_init[i].bandwidth = float(est[i, 0]), float(est[i, 1])
        p.mfcc_coeffs, p.mfcc_lo_hz, p.mfcc_hi_hz = (mfcc if mfcc else (0, 0.0, 0.0))
        return p

    def columns(self):
        """name -> (first column, width) of the per-frame record."""
        cols, c = {"pitch": (0, 2)}, 2
        if self.formant_order:
            cols["formants"] = (c, 2 * int(self.n_est)); c += 2 * int(self.n_est)
        if self.mfcc_coeffs:
            cols["mfcc"] = (c, int(self.mfcc_coeffs)); c += int(self.mfcc_coeffs)
        if self.lpc_order:
            cols["lpc"] = (c, int(self.lpc_order) + 1); c += int(self.lpc_order) + 1
        return cols


_lib = None


def exported_symbols():
    """Function names declared in include/voxbox_hip.h (the ABI the library must export)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vbx_[a-z0-9_]+)\s*\(", text)))


def load_library():
    """Loads libvoxbox_hip.so.  Fails loudly if it was not built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VoxBoxError(
            f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (or __graft_entry__.build()); "
            "there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, sz, dbl, i32, i64 = C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_int64
    sig = {
        "vbx_abi_version": (C.c_int, []),
        "vbx_ctx_create": (C.c_int, [C.POINTER(vp), i32, vp]),
        "vbx_ctx_destroy": (None, [vp]),
        "vbx_sync": (C.c_int, [vp]),
        "vbx_last_error": (C.c_char_p, [vp]),
        "vbx_device_info": (C.c_int, [vp, C.c_char_p, sz, C.POINTER(C.c_int)]),
        "vbx_malloc": (C.c_int, [vp, C.POINTER(vp), sz]),
        "vbx_free": (C.c_int, [vp, vp]),
        "vbx_memcpy_h2d": (C.c_int, [vp, vp, vp, sz]),
        "vbx_memcpy_d2h": (C.c_int, [vp, vp, vp, sz]),
        "vbx_memset": (C.c_int, [vp, vp, i32, sz]),
        "vbx_timer_begin": (C.c_int, [vp]),
        "vbx_timer_end": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "vbx_profile_enable": (C.c_int, [vp, i32]),
        "vbx_profile_reset": (C.c_int, [vp]),
        "vbx_profile_get": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
        "vbx_profile_stream": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_int)]),
        "vbx_profile_names": (C.c_int, [vp, C.c_char_p, sz]),
        "vbx_profile_pitch_work": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "vbx_window_table_f64": (C.c_int, [i32, sz, vp]),
        "vbx_frame_count": (sz, [sz, sz, sz]),
        "vbx_hz_to_mel": (dbl, [dbl]),
        "vbx_mel_to_hz": (dbl, [dbl]),
        "vbx_find_formants_real_work_size": (sz, [sz, sz]),
        "vbx_find_formants_complex_work_size": (sz, [sz]),
        "vbx_autocorrelate_f64": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, vp]),
        "vbx_normalize_f64": (C.c_int, [vp, vp, sz, sz]),
        "vbx_interpolate_sinc_f64": (C.c_int, [vp, vp, sz, C.c_long, sz, vp, sz, sz, vp, vp]),
        "vbx_improve_extremum_f64": (C.c_int, [vp, vp, sz, C.c_long, sz, vp, sz, sz, vp, vp]),
        "vbx_improve_extremum_ex_f64": (C.c_int, [vp, vp, sz, C.c_long, sz, vp, sz, i32, sz, i32, vp, vp]),
        "vbx_pitch_f64": (C.c_int, [vp, vp, sz, sz, sz, vp, dbl, dbl, dbl, dbl, sz, vp, vp, vp]),
        "vbx_lpc_f64": (C.c_int, [vp, vp, sz, sz, sz, vp]),
        "vbx_lpc_mut_f64": (C.c_int, [vp, vp, sz, sz, sz, vp, vp]),
        "vbx_window_table_f32": (C.c_int, [i32, sz, vp]),
        "vbx_autocorrelate_f32": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, vp]),
        "vbx_autocorrelate_f32_wide": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, vp]),
        "vbx_normalize_f32": (C.c_int, [vp, vp, sz, sz]),
        "vbx_lpc_mut_f32": (C.c_int, [vp, vp, sz, sz, sz, vp, vp]),
        "vbx_lpc_mut_f32_wide": (C.c_int, [vp, vp, sz, sz, sz, vp, vp]),
        "vbx_autocorr_lpc_f32": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, i32, vp, vp]),
        "vbx_autocorr_lpc_f32_wide": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, i32, vp, vp]),
        "vbx_lpc_burg_f32": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, vp, vp]),
        "vbx_lpc_burg_f32_wide": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, vp, vp]),
        "vbx_mfcc_f32": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, dbl, dbl, dbl, vp, vp]),
        "vbx_pitch_f32": (C.c_int, [vp, vp, sz, sz, sz, vp, C.c_float, C.c_float, C.c_float, C.c_float, sz, vp, vp, vp]),
        "vbx_pitch_f32_wide": (C.c_int, [vp, vp, sz, sz, sz, vp, C.c_float, C.c_float, C.c_float, C.c_float, sz, vp, vp, vp]),
        "vbx_autocorr_lpc_f64": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, i32, vp, vp]),
        "vbx_lpc_burg_f64": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, vp, vp]),
        "vbx_find_roots_c64": (C.c_int, [vp, vp, sz, sz, vp]),
        "vbx_laguerre_c64": (C.c_int, [vp, vp, sz, sz, _Complex, vp]),
        "vbx_div_polynomial_c64": (C.c_int, [vp, vp, vp, sz, sz, vp, vp]),
        "vbx_degree_c64": (sz, [vp, sz]),
        "vbx_off_low_c64": (sz, [vp, sz]),
        "vbx_ring_frames_f64": (C.c_int, [vp, vp, sz, sz, sz, sz, sz, vp]),
        "vbx_find_roots_c32": (C.c_int, [vp, vp, sz, sz, vp]),
        "vbx_laguerre_c32": (C.c_int, [vp, vp, sz, sz, _Complex32, vp]),
        "vbx_to_resonance_c64": (C.c_int, [vp, vp, sz, sz, dbl, vp, vp]),
        "vbx_estimate_formants_f64": (C.c_int, [vp, vp, sz, sz, vp, sz, vp, sz, vp, vp]),
        "vbx_find_formants_f64": (C.c_int, [vp, vp, sz, sz, sz, dbl, sz, vp, sz, vp, sz, vp, vp, vp, vp, vp]),
        "vbx_mfcc_f64": (C.c_int, [vp, vp, sz, sz, sz, vp, sz, dbl, dbl, dbl, vp, vp]),
        "vbx_dct_f64": (C.c_int, [vp, vp, sz, sz, vp]),
        "vbx_mfcc_bins": (C.c_int, [sz, sz, dbl, dbl, dbl, vp]),
        "vbx_resampled_len": (sz, [sz, dbl]),
        "vbx_resample_linear_f64": (C.c_int, [vp, vp, sz, sz, sz, dbl, vp]),
        "vbx_pcm16_to_f64": (C.c_int, [vp, vp, sz, vp]),
        "vbx_rms_f64": (C.c_int, [vp, vp, sz, sz, sz, vp, vp]),
        "vbx_preemphasis_f64": (C.c_int, [vp, vp, sz, sz, sz, dbl, vp]),
        "vbx_synth_speech_f64": (C.c_int, [vp, vp, sz, C.c_uint64, dbl, C.c_uint64]),
        "vbx_selftest_lanes": (C.c_int, [vp, vp]),
        "vbx_internal_last_unsure_count": (C.c_int, [vp, vp]),
        "vbx_internal_last_burg_direct_count": (C.c_int, [vp, vp]),
        "vbx_internal_last_lpc_exact_count": (C.c_int, [vp, vp]),
        "vbx_internal_estimate_formants_counted": (C.c_int, [vp, vp, sz, sz, vp, vp, sz, vp, sz, vp, vp]),
        "vbx_internal_last_roots_direct_count": (C.c_int, [vp, vp]),
        "vbx_internal_last_spectral_split": (C.c_int, [vp]),
        "vbx_internal_last_mfcc_interp": (C.c_int, [vp]),
        "vbx_record_doubles": (sz, [C.POINTER(AnalysisParams)]),
        "vbx_analyze_frames_f64": (C.c_int, [vp, vp, sz, sz, sz, C.POINTER(AnalysisParams), vp, sz, vp, sz, vp]),
        "vbx_analyze_frames_pcm16": (C.c_int, [vp, vp, sz, sz, sz, C.POINTER(AnalysisParams), vp, sz, vp, sz, vp]),
        "vbx_shard_range": (C.c_int, [sz, i32, i32, vp, sz, C.POINTER(sz), C.POINTER(sz)]),
        "vbx_shard_samples": (C.c_int, [sz, sz, sz, sz, C.POINTER(sz), C.POINTER(sz)]),
        "vbx_shard_plan": (C.c_int, [sz, i32, i32, vp, sz, C.POINTER(ShardPlan)]),
        "vbx_shard_local_segments": (C.c_int, [C.POINTER(ShardPlan), vp, sz, vp, sz, C.POINTER(sz)]),
        "vbx_track_stitch_f64": (C.c_int, [vp, vp, sz, sz, sz, sz, vp, vp]),
        "vbx_comm_stitch_tracks_f64": (C.c_int, [vp, vp, vp, sz, sz, C.POINTER(ShardPlan), vp, i32]),
        "vbx_comm_unique_id": (C.c_int, [vp]),
        "vbx_comm_create": (C.c_int, [vp, vp, i32, i32, C.POINTER(vp)]),
        "vbx_comm_destroy": (None, [vp]),
        "vbx_gather_records_f64": (C.c_int, [vp, vp, vp, vp, sz, i32, vp, i32]),
        "vbx_gather_plan": (C.c_int, [vp, i32, i32, i32, sz, vp, vp, vp]),
        "vbx_comm_live_count": (C.c_int, []),
        "vbx_comm_wait": (C.c_int, [vp, vp, i32]),
        "vbx_comm_sync": (C.c_int, [vp]),
        "vbx_comm_selftest": (C.c_int, [vp, vp, sz]),
    }
    for name, (res, args) in sig.items():
        if name.startswith("vbx_internal_") and os.environ.get("VBX_LIB_PATH") and not hasattr(L, name):
            continue                   # an experiment build of an earlier round (tools/experiments/bitcompare_libs.py): probes it does not have
        fn = getattr(L, name)          # AttributeError here = symbol missing from the build
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# ---- host-only helpers (no GPU needed) --------------------------------------------------

def poly_degree(poly):
    """Polynomial::degree (src/polynomial.rs:26-28) of one host polynomial."""
    p = np.ascontiguousarray(poly, dtype=np.complex128)
    return int(load_library().vbx_degree_c64(p.ctypes.data, p.size))


def poly_off_low(poly):
    """Polynomial::off_low (src/polynomial.rs:30-32)."""
    p = np.ascontiguousarray(poly, dtype=np.complex128)
    return int(load_library().vbx_off_low_c64(p.ctypes.data, p.size))


def window_table(kind, n):
    out = np.empty(n, dtype=np.float64)
    rc = load_library().vbx_window_table_f64(kind, n, out.ctypes.data)
    if rc != 0:
        raise VoxBoxError("vbx_window_table_f64 failed")
    return out


def frame_count(n_samples, frame_len, hop):
    return int(load_library().vbx_frame_count(n_samples, frame_len, hop))


def hz_to_mel(hz):
    return load_library().vbx_hz_to_mel(hz)


def mel_to_hz(mel):
    return load_library().vbx_mel_to_hz(mel)


def mfcc_bins(frame_len, num_coeffs, lo_hz, hi_hz, sample_rate):
    """vbx_mfcc_bins: (bins[num_coeffs + 2], panics) -- the mel filter bank's bins (src/spectrum.rs:411-414) and whether the
    reference panics on every frame of this geometry."""
    out = np.zeros(num_coeffs + 2, dtype=np.int32)
    rc = load_library().vbx_mfcc_bins(frame_len, num_coeffs, lo_hz, hi_hz, sample_rate, out.ctypes.data)
    if rc < 0:
        raise VoxBoxError("vbx_mfcc_bins: bad argument")
    return out, bool(rc)


def shard_range(n_frames, world, rank, seg_start=None):
    """vbx_shard_range: frames [lo, hi) of `rank` -- the EVEN split.  With seg_start a cut moves to an utterance start only
    when one lies within 1/32 of a shard after it; otherwise it stays INSIDE the utterance (ABI 4; up to ABI 3 every cut
    moved to a boundary).  A caller that analyses [lo, hi) per rank on its own therefore restarts the tracker in the middle
    of an utterance: use shard_plan + shard_local_segments + Comm.stitch_tracks ("From ABI 3 to ABI 4"),
    or shard.segment_aligned_ranges for cuts on boundaries only."""
    lo, hi = C.c_size_t(), C.c_size_t()
    seg = None if seg_start is None else np.ascontiguousarray(seg_start, dtype=np.int64)
    rc = load_library().vbx_shard_range(n_frames, world, rank, None if seg is None else seg.ctypes.data,
                                        0 if seg is None else seg.size, C.byref(lo), C.byref(hi))
    if rc != 0:
        raise VoxBoxError("vbx_shard_range: bad argument")
    return lo.value, hi.value


def shard_plan(n_frames, world, rank, seg_start=None):
    """vbx_shard_plan: this rank's frames [lo, hi), the warm-up frames before them and whether its track continues from the
    previous rank / into the next one."""
    plan = ShardPlan()
    seg = None if seg_start is None else np.ascontiguousarray(seg_start, dtype=np.int64)
    rc = load_library().vbx_shard_plan(n_frames, world, rank, None if seg is None else seg.ctypes.data,
                                       0 if seg is None else seg.size, C.byref(plan))
    if rc != 0:
        raise VoxBoxError("vbx_shard_plan: bad argument")
    return plan


def shard_local_segments(plan, seg_start=None):
    """vbx_shard_local_segments: utterance starts of the frames [lo - warm, hi), re-based to the shard."""
    seg = None if seg_start is None else np.ascontiguousarray(seg_start, dtype=np.int64)
    n = C.c_size_t()
    L = load_library()
    sp, sn = (None if seg is None else seg.ctypes.data), (0 if seg is None else seg.size)
    if L.vbx_shard_local_segments(C.byref(plan), sp, sn, None, 0, C.byref(n)) != 0:
        raise VoxBoxError("vbx_shard_local_segments: bad argument")
    out = np.zeros(n.value, dtype=np.int64)
    if L.vbx_shard_local_segments(C.byref(plan), sp, sn, out.ctypes.data, out.size, C.byref(n)) != 0:
        raise VoxBoxError("vbx_shard_local_segments: bad argument")
    return out


def shard_samples(lo, hi, frame_len, hop):
    s0, s1 = C.c_size_t(), C.c_size_t()
    if load_library().vbx_shard_samples(lo, hi, frame_len, hop, C.byref(s0), C.byref(s1)) != 0:
        raise VoxBoxError("vbx_shard_samples: bad argument")
    return s0.value, s1.value


GATHER_NONE, GATHER_RECV, GATHER_SEND, GATHER_COPY = 0, 1, 2, 3


def gather_plan(rows, rank, dst, row_doubles):
    """vbx_gather_plan: (offset[world], count[world], op[world]) of the record gather as `rank` sees it (host only)."""
    r = np.ascontiguousarray(rows, dtype=np.int64)
    off, cnt, op = np.zeros(r.size, np.int64), np.zeros(r.size, np.int64), np.zeros(r.size, np.int32)
    rc = load_library().vbx_gather_plan(r.ctypes.data, r.size, rank, dst, row_doubles, off.ctypes.data, cnt.ctypes.data, op.ctypes.data)
    if rc != 0:
        raise VoxBoxError("vbx_gather_plan: bad argument")
    return off, cnt, op


def comm_live_count():
    return int(load_library().vbx_comm_live_count())


def comm_unique_id():
    """ncclGetUniqueId through the library (rank 0); ship the bytes to the other ranks."""
    buf = C.create_string_buffer(128)
    if load_library().vbx_comm_unique_id(buf) != 0:
        msg = load_library().vbx_last_error(None)
        raise VoxBoxError(f"vbx_comm_unique_id failed: {msg.decode() if msg else ''}")
    return buf.raw


class Comm:
    """vbx_comm: this rank's RCCL communicator for the record gather (one per VoxBox context)."""

    def __init__(self, vb, unique_id, world, rank):
        self.vb, self.world, self.rank = vb, world, rank
        h = C.c_void_p()
        vb._check(vb.L.vbx_comm_create(vb.ctx, C.c_char_p(unique_id), world, rank, C.byref(h)))
        self.h = h

    def gather_records(self, local, rows, row_doubles, dst=0, out=None, slot=0):
        r = np.ascontiguousarray(rows, dtype=np.int64)
        self.vb._check(self.vb.L.vbx_gather_records_f64(self.vb.ctx, self.h, _ptr(local), r.ctypes.data, row_doubles, dst,
                                                        _ptr(out), slot))

    def stitch_tracks(self, formants, n_frames, formants_ld, plan, changed=None, slot=0):
        """vbx_comm_stitch_tracks_f64: the formant rows of the last analyze / find_formants call, continued from the previous
        rank's last row (and this rank's last row passed on), on the communicator's stream."""
        self.vb._check(self.vb.L.vbx_comm_stitch_tracks_f64(self.vb.ctx, self.h, _ptr(formants), n_frames, formants_ld,
                                                            C.byref(plan), _ptr(changed), slot))

    def wait(self, slot):
        self.vb._check(self.vb.L.vbx_comm_wait(self.vb.ctx, self.h, slot))

    def sync(self):
        self.vb._check(self.vb.L.vbx_comm_sync(self.h))

    def selftest(self, n=1 << 16):
        self.vb._check(self.vb.L.vbx_comm_selftest(self.vb.ctx, self.h, n))

    def close(self):
        if getattr(self, "h", None):
            self.vb.L.vbx_comm_destroy(self.h)
            self.h = None


class DeviceArray:
    """A device allocation owned by a VoxBox context (freed with it or via .free())."""

    def __init__(self, vb, nbytes, dtype=np.float64, shape=None):
        self.vb, self.nbytes, self.dtype = vb, int(nbytes), np.dtype(dtype)
        self.shape = shape if shape is not None else (self.nbytes // self.dtype.itemsize,)
        p = C.c_void_p()
        vb._check(vb.L.vbx_malloc(vb.ctx, C.byref(p), self.nbytes))
        self.ptr = p.value
        vb._allocs.add(self)

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        self.vb._check(self.vb.L.vbx_memcpy_d2h(self.vb.ctx, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def numpy_slice(self, start, count):
        """`count` elements from flat element index `start` (a partial download of a large buffer)."""
        out = np.empty(int(count), dtype=self.dtype)
        self.vb._check(self.vb.L.vbx_memcpy_d2h(self.vb.ctx, out.ctypes.data, self.ptr + int(start) * self.dtype.itemsize, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.vb.L.vbx_free(self.vb.ctx, self.ptr)
            self.ptr = None
            self.vb._allocs.discard(self)


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return a.ptr
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):          # torch tensor on the device
        return a.data_ptr()
    raise TypeError(f"not a device buffer: {type(a)}")


class VoxBox:
    """One libvoxbox_hip context = one GPU + one HIP stream."""

    def __init__(self, device=0, stream=None):
        self.L = load_library()
        self._allocs = set()
        ctx = C.c_void_p()
        rc = self.L.vbx_ctx_create(C.byref(ctx), device, stream)
        if rc != 0:
            msg = self.L.vbx_last_error(None)
            raise VoxBoxError(f"vbx_ctx_create failed ({rc}): {msg.decode() if msg else ''}")
        self.ctx = ctx
        self._tables = {}

    # -- plumbing ---------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self.L.vbx_last_error(self.ctx)
            raise VoxBoxError(f"libvoxbox_hip error {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "ctx", None):
            for a in list(self._allocs):
                a.free()
            self.L.vbx_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        self._check(self.L.vbx_sync(self.ctx))

    def device_info(self):
        buf = C.create_string_buffer(128)
        cu = C.c_int()
        self._check(self.L.vbx_device_info(self.ctx, buf, 128, C.byref(cu)))
        return buf.value.decode(), cu.value

    def empty(self, shape, dtype=np.float64):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        n = int(np.prod(shape)) if shape else 1
        return DeviceArray(self, max(n, 1) * np.dtype(dtype).itemsize, dtype, shape)

    def to_device(self, a, dtype=None):
        a = np.ascontiguousarray(a, dtype=dtype)
        d = DeviceArray(self, max(a.nbytes, 16), a.dtype, a.shape)
        if a.nbytes:
            self._check(self.L.vbx_memcpy_h2d(self.ctx, d.ptr, a.ctypes.data, a.nbytes))
        return d

    def zeros(self, shape, dtype=np.float64):
        d = self.empty(shape, dtype)
        self._check(self.L.vbx_memset(self.ctx, d.ptr, 0, d.nbytes))
        return d

    def window(self, kind, n):
        """device copy of a host-built window table (cached)."""
        key = (kind, n)
        if key not in self._tables:
            self._tables[key] = self.to_device(window_table(kind, n))
        return self._tables[key]

    def timer_begin(self):
        self._check(self.L.vbx_timer_begin(self.ctx))

    def timer_end(self):
        ms = C.c_float()
        self._check(self.L.vbx_timer_end(self.ctx, C.byref(ms)))
        return ms.value

    def profile(self, on=True):
        self._check(self.L.vbx_profile_enable(self.ctx, 1 if on else 0))

    def profile_reset(self):
        self._check(self.L.vbx_profile_reset(self.ctx))

    def profile_pitch_work(self):
        """(frames, candidates, sinc evaluations, sinc terms) the pitch refine kernel executed while profiling."""
        w = (C.c_uint64 * 4)()
        self._check(self.L.vbx_profile_pitch_work(self.ctx, w))
        return tuple(int(v) for v in w)

    def profile_report(self):
        buf = C.create_string_buffer(4096)
        self._check(self.L.vbx_profile_names(self.ctx, buf, 4096))
        out = {}
        for name in buf.value.decode().split("\n"):
            if not name:
                continue
            ms, cnt = C.c_double(), C.c_long()
            self._check(self.L.vbx_profile_get(self.ctx, name.encode(), C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    def profile_streams(self):
        """name -> 0 (the context's stream: the critical path), 1 (side stream), 2 (tracker time slices)."""
        out = {}
        for name in self.profile_report():
            sid = C.c_int()
            self._check(self.L.vbx_profile_stream(self.ctx, name.encode(), C.byref(sid)))
            out[name] = sid.value
        return out

    def _frames(self, x, frame_len=None, stride=None, n_frames=None):
        """Resolves a frame batch: 2-D host array = dense [F, N]; 1-D host array + (frame_len,
        stride) = Windower view; device buffers need all of frame_len/stride/n_frames."""
        tmp = None
        if isinstance(x, np.ndarray):
            if x.ndim == 2:
                n_frames, frame_len = x.shape
                stride = frame_len
            else:
                assert frame_len and stride
                n_frames = frame_count(x.size, frame_len, stride) if n_frames is None else n_frames
            tmp = self.to_device(x, np.float64)
            ptr = tmp.ptr
        else:
            assert frame_len and stride and n_frames is not None
            ptr = _ptr(x)
        return ptr, int(n_frames), int(frame_len), int(stride), tmp

    # -- periodic.rs ------------------------------------------------------------------
    def autocorrelate(self, x, n_lags, frame_len=None, stride=None, n_frames=None, window=None, out=None):
        ptr, F, N, S, tmp = self._frames(x, frame_len, stride, n_frames)
        o = out if out is not None else self.empty((F, n_lags))
        self._check(self.L.vbx_autocorrelate_f64(self.ctx, ptr, F, N, S, _ptr(window), n_lags, _ptr(o)))
        return self._finish(o, out, tmp)

    def normalize(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        d = self.to_device(rows)
        self._check(self.L.vbx_normalize_f64(self.ctx, d.ptr, rows.shape[0], rows.shape[1]))
        r = d.numpy()
        d.free()
        return r

    def interpolate_sinc(self, y, offset, nx, xs, depth):
        y = np.ascontiguousarray(y, dtype=np.float64)
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        dy, dx = self.to_device(y), self.to_device(xs)
        o, st = self.empty(xs.size), self.empty(xs.size, np.int32)
        self._check(self.L.vbx_interpolate_sinc_f64(self.ctx, dy.ptr, y.size, offset, nx, dx.ptr, xs.size, depth, o.ptr, st.ptr))
        r = (o.numpy(), st.numpy())
        for d in (dy, dx, o, st):
            d.free()
        return r

    def improve_extremum(self, y, offset, nx, ixmid, depth):
        y = np.ascontiguousarray(y, dtype=np.float64)
        ix = np.ascontiguousarray(ixmid, dtype=np.float64)
        dy, dx = self.to_device(y), self.to_device(ix)
        o, st = self.empty((ix.size, 2)), self.empty(ix.size, np.int32)
        self._check(self.L.vbx_improve_extremum_f64(self.ctx, dy.ptr, y.size, offset, nx, dx.ptr, ix.size, depth, o.ptr, st.ptr))
        r = (o.numpy(), st.numpy())
        for d in (dy, dx, o, st):
            d.free()
        return r

    def improve_extremum_ex(self, y, offset, nx, ixmid, interpolation, depth=0, is_max=True):
        """improve_extremum with any Interpolation arm (0 None, 1 Parabolic, 2 Sinc(depth)) and is_max."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        ix = np.ascontiguousarray(ixmid, dtype=np.float64)
        dy, dx = self.to_device(y), self.to_device(ix)
        o, st = self.empty((ix.size, 2)), self.empty(ix.size, np.int32)
        self._check(self.L.vbx_improve_extremum_ex_f64(self.ctx, dy.ptr, y.size, offset, nx, dx.ptr, ix.size, interpolation, depth,
                                                       1 if is_max else 0, o.ptr, st.ptr))
        r = (o.numpy(), st.numpy())
        for d in (dy, dx, o, st):
            d.free()
        return r

    def pitch(self, x, sample_rate, threshold, fmin, fmax, kmax=8, frame_len=None, stride=None, n_frames=None,
              window=None, out=None):
        """Returns (cand[F, kmax, 2], count[F], status[F]) as numpy, or writes into `out`
        = (cand, count, status) device buffers and returns None.  kmax = pitch_max_candidates(frame_len) returns
        the reference's whole Vec for every frame."""
        ptr, F, N, S, tmp = self._frames(x, frame_len, stride, n_frames)
        if out is None:
            cand, cnt, st = self.empty((F, kmax, 2)), self.empty(F, np.int32), self.empty(F, np.int32)
        else:
            cand, cnt, st = out
        self._check(self.L.vbx_pitch_f64(self.ctx, ptr, F, N, S, _ptr(window), sample_rate, threshold, fmin, fmax,
                                         kmax, _ptr(cand), _ptr(cnt), _ptr(st)))
        if out is not None:
            return None
        r = (cand.numpy(), cnt.numpy(), st.numpy())
        for d in (cand, cnt, st, tmp):
            if d is not None:
                d.free()
        return r

    # -- spectrum.rs: LPC -------------------------------------------------------------
    def lpc(self, r, n_coeffs):
        r = np.ascontiguousarray(r, dtype=np.float64)
        d = self.to_device(r)
        o = self.empty((r.shape[0], n_coeffs + 1))
        self._check(self.L.vbx_lpc_f64(self.ctx, d.ptr, r.shape[0], r.shape[1], n_coeffs, o.ptr))
        res = o.numpy()
        d.free(); o.free()
        return res

    def lpc_mut(self, r, n_coeffs):
        """LPC::lpc_mut: returns (ac[F, n_coeffs + 1], kc[F, n_coeffs]) -- coefficients and reflection coefficients."""
        r = np.ascontiguousarray(r, dtype=np.float64)
        d = self.to_device(r)
        o, k = self.empty((r.shape[0], n_coeffs + 1)), self.empty((r.shape[0], n_coeffs))
        self._check(self.L.vbx_lpc_mut_f64(self.ctx, d.ptr, r.shape[0], r.shape[1], n_coeffs, o.ptr, k.ptr))
        res = (o.numpy(), k.numpy())
        d.free(); o.free(); k.free()
        return res

    def autocorr_lpc(self, x, n_coeffs, normalize=False, frame_len=None, stride=None, n_frames=None, window=None,
                     out=None):
        ptr, F, N, S, tmp = self._frames(x, frame_len, stride, n_frames)
        if out is None:
            r, a = self.empty((F, n_coeffs + 1)), self.empty((F, n_coeffs + 1))
        else:
            r, a = out
        self._check(self.L.vbx_autocorr_lpc_f64(self.ctx, ptr, F, N, S, _ptr(window), n_coeffs, 1 if normalize else 0,
                                                _ptr(r), _ptr(a)))
        if out is not None:
            return None
        res = (r.numpy(), a.numpy())
        for d in (r, a, tmp):
            if d is not None:
                d.free()
        return res

    def lpc_praat(self, x, n_coeffs, frame_len=None, stride=None, n_frames=None, window=None, out=None):
        """Burg (LPC::lpc_praat).  Returns (coeffs[F, p], status[F])."""
        ptr, F, N, S, tmp = self._frames(x, frame_len, stride, n_frames)
        if out is None:
            co, st = self.empty((F, n_coeffs)), self.empty(F, np.int32)
        else:
            co, st = out
        self._check(self.L.vbx_lpc_burg_f64(self.ctx, ptr, F, N, S, _ptr(window), n_coeffs, _ptr(co), _ptr(st)))
        if out is not None:
            return None
        res = (co.numpy(), st.numpy())
        for d in (co, st, tmp):
            if d is not None:
                d.free()
        return res

    # -- polynomial.rs ----------------------------------------------------------------
    def find_roots(self, polys):
        """polys: [F, len] complex (coefficient of x^j at index j).  Returns (roots[F, len], status[F]);
        row layout as find_roots_mut leaves `self`: roots in discovery order, then zeros."""
        p = np.ascontiguousarray(polys, dtype=np.complex128)
        d = self.to_device(p)
        st = self.empty(p.shape[0], np.int32)
        self._check(self.L.vbx_find_roots_c64(self.ctx, d.ptr, p.shape[0], p.shape[1], st.ptr))
        res = (d.numpy(), st.numpy())
        d.free(); st.free()
        return res

    def laguerre(self, polys, start):
        p = np.ascontiguousarray(polys, dtype=np.complex128)
        d = self.to_device(p)
        o = self.empty(p.shape[0], np.complex128)
        self._check(self.L.vbx_laguerre_c64(self.ctx, d.ptr, p.shape[0], p.shape[1], _Complex(start.real, start.imag), o.ptr))
        res = o.numpy()
        d.free(); o.free()
        return res

    def div_polynomial(self, polys, others):
        """div_polynomial_mut (src/polynomial.rs:155-195) per row: returns (quotient rows, remainder rows, status)."""
        p = np.ascontiguousarray(polys, dtype=np.complex128)
        o = np.ascontiguousarray(others, dtype=np.complex128)
        d, do = self.to_device(p), self.to_device(o)
        rem, st = self.empty(p.shape, np.complex128), self.empty(p.shape[0], np.int32)
        self._check(self.L.vbx_div_polynomial_c64(self.ctx, d.ptr, do.ptr, p.shape[0], p.shape[1], rem.ptr, st.ptr))
        res = (d.numpy(), rem.numpy(), st.numpy())
        for b in (d, do, rem, st):
            b.free()
        return res

    def ring_frames(self, ring, head, n_frames, frame_len, stride):
        """VecDeque view -> dense [F, frame_len] DeviceArray (src/periodic.rs:291-304)."""
        r = ring if isinstance(ring, DeviceArray) else self.to_device(np.ascontiguousarray(ring, dtype=np.float64))
        cap = int(np.prod(r.shape))
        out = self.empty((n_frames, frame_len))
        self._check(self.L.vbx_ring_frames_f64(self.ctx, r.ptr, cap, head, n_frames, frame_len, stride, out.ptr))
        if r is not ring:
            self.sync(); r.free()
        return out

    def find_roots_f32(self, polys):
        """The Complex<f32> instantiation of find_roots (src/polynomial.rs:336-386): polys [F, len] complex64."""
        p = np.ascontiguousarray(polys, dtype=np.complex64)
        d = self.to_device(p)
        st = self.empty(p.shape[0], np.int32)
        self._check(self.L.vbx_find_roots_c32(self.ctx, d.ptr, p.shape[0], p.shape[1], st.ptr))
        res = (d.numpy(), st.numpy())
        d.free(); st.free()
        return res

    def laguerre_f32(self, polys, start):
        p = np.ascontiguousarray(polys, dtype=np.complex64)
        d = self.to_device(p)
        o = self.empty(p.shape[0], np.complex64)
        self._check(self.L.vbx_laguerre_c32(self.ctx, d.ptr, p.shape[0], p.shape[1], _Complex32(start.real, start.imag), o.ptr))
        res = o.numpy()
        d.free(); o.free()
        return res

    # -- spectrum.rs: resonances / tracker ---------------------------------------------
    def to_resonance(self, roots, sample_rate):
        r = np.ascontiguousarray(roots, dtype=np.complex128)
        d = self.to_device(r)
        o, c = self.empty((r.shape[0], r.shape[1], 2)), self.empty(r.shape[0], np.int32)
        self._check(self.L.vbx_to_resonance_c64(self.ctx, d.ptr, r.shape[0], r.shape[1], sample_rate, o.ptr, c.ptr))
        res = (o.numpy(), c.numpy())
        for b in (d, o, c):
            b.free()
        return res

    def estimate_formants(self, res, est_init, seg_start=None, frame_status=None, res_count=None):
        """FormantExtractor over [F, n_res, 2] resonance rows; returns estimates [F, n_est, 2].  res_count (test probe): the
        per-row counts find_formants keeps beside its rows -- the scan may then take the tracker's index form."""
        r = np.ascontiguousarray(res, dtype=np.float64)
        e = np.ascontiguousarray(est_init, dtype=np.float64)
        F, n_res, n_est = r.shape[0], r.shape[1], e.shape[0]
        d = self.to_device(r)
        o = self.empty((F, n_est, 2))
        seg = None if seg_start is None else np.ascontiguousarray(seg_start, dtype=np.int64)
        fs = None if frame_status is None else self.to_device(np.ascontiguousarray(frame_status, dtype=np.int32))
        rc_d = None if res_count is None else self.to_device(np.ascontiguousarray(res_count, dtype=np.int32))
        if rc_d is not None:
            self._check(self.L.vbx_internal_estimate_formants_counted(
                self.ctx, d.ptr, F, n_res, rc_d.ptr, None if seg is None else seg.ctypes.data, 0 if seg is None else seg.size,
                e.ctypes.data, n_est, _ptr(fs), o.ptr))
        else:
            self._check(self.L.vbx_estimate_formants_f64(
                self.ctx, d.ptr, F, n_res, None if seg is None else seg.ctypes.data, 0 if seg is None else seg.size,
                e.ctypes.data, n_est, _ptr(fs), o.ptr))
        out = o.numpy()
        for b in (d, o, fs, rc_d):
            if b is not None:
                b.free()
        return out

    def find_formants(self, x, sample_rate, n_coeffs, est_init, seg_start=None, frame_len=None, stride=None,
                      n_frames=None, out=None, want=("formants", "res", "count", "coeffs", "status")):
        """vox_box::find_formants over a batch (resample_ratio = 1.0).  est_init: [n_est, 2]."""
        ptr, F, N, S, tmp = self._frames(x, frame_len, stride, n_frames)
        e = np.ascontiguousarray(est_init, dtype=np.float64)
        n_est = e.shape[0]
        seg = None if seg_start is None else np.ascontiguousarray(seg_start, dtype=np.int64)
        if out is None:
            bufs = {
                "formants": self.empty((F, n_est, 2)),
                "res": self.empty((F, MAX_RESONANCES, 2)) if "res" in want else None,
                "count": self.empty(F, np.int32) if "count" in want else None,
                "coeffs": self.empty((F, n_coeffs)) if "coeffs" in want else None,
                "status": self.empty(F, np.int32) if "status" in want else None,
            }
        else:
            bufs = out
        self._check(self.L.vbx_find_formants_f64(
            self.ctx, ptr, F, N, S, sample_rate, n_coeffs,
            None if seg is None else seg.ctypes.data, 0 if seg is None else seg.size,
            e.ctypes.data, n_est, _ptr(bufs["formants"]), _ptr(bufs.get("res")), _ptr(bufs.get("count")),
            _ptr(bufs.get("coeffs")), _ptr(bufs.get("status"))))
        if out is not None:
            return None
        res = {k: (v.numpy() if v is not None else None) for k, v in bufs.items()}
        for v in list(bufs.values()) + [tmp]:
            if v is not None:
                v.free()
        return res

    def track_stitch(self, formants, n_frames, formants_ld, first, stop, state_in, changed=None):
        """vbx_track_stitch_f64: rows [first, stop) of the LAST find_formants / analyze_frames call's formant tracks, corrected
        to follow from `state_in` (device pointer: the row the previous shard ends with)."""
        self._check(self.L.vbx_track_stitch_f64(self.ctx, _ptr(formants), n_frames, formants_ld, first, stop, _ptr(state_in),
                                                _ptr(changed)))

    # -- Sample = f32: float frames in, float results out ---------------
    def _frames32(self, x):
        """Dense [F, N] float32 host batch -> device."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2
        return self.to_device(x, np.float32), x.shape[0], x.shape[1]

    def _win32(self, window):
        return None if window is None else self.to_device(np.ascontiguousarray(window, dtype=np.float32), np.float32)

    def autocorrelate_f32(self, x, n_lags, window=None, wide=False):
        """wide=False: the reference-faithful f32 folds; wide=True: f64 arithmetic on the widened frame, rounded once."""
        d, F, N = self._frames32(x)
        w = self._win32(window)
        o = self.empty((F, n_lags), np.float32)
        self._check((self.L.vbx_autocorrelate_f32_wide if wide else self.L.vbx_autocorrelate_f32)(self.ctx, d.ptr, F, N, N, _ptr(w), n_lags, o.ptr))
        r = o.numpy()
        for b in (d, w, o):
            if b is not None:
                b.free()
        return r

    def normalize_f32(self, rows):
        d, F, N = self._frames32(rows)
        self._check(self.L.vbx_normalize_f32(self.ctx, d.ptr, F, N))
        r = d.numpy()
        d.free()
        return r

    def lpc_mut_f32(self, r, n_coeffs, wide=False):
        d, F, N = self._frames32(r)
        o, k = self.empty((F, n_coeffs + 1), np.float32), self.empty((F, n_coeffs), np.float32)
        self._check((self.L.vbx_lpc_mut_f32_wide if wide else self.L.vbx_lpc_mut_f32)(self.ctx, d.ptr, F, N, n_coeffs, o.ptr, k.ptr))
        res = (o.numpy(), k.numpy())
        for b in (d, o, k):
            b.free()
        return res

    def autocorr_lpc_f32(self, x, n_coeffs, normalize=False, window=None, wide=False):
        d, F, N = self._frames32(x)
        w = self._win32(window)
        r, a = self.empty((F, n_coeffs + 1), np.float32), self.empty((F, n_coeffs + 1), np.float32)
        self._check((self.L.vbx_autocorr_lpc_f32_wide if wide else self.L.vbx_autocorr_lpc_f32)(self.ctx, d.ptr, F, N, N, _ptr(w), n_coeffs, int(bool(normalize)), r.ptr, a.ptr))
        res = (r.numpy(), a.numpy())
        for b in (d, w, r, a):
            if b is not None:
                b.free()
        return res

    def lpc_praat_f32(self, x, n_coeffs, window=None, wide=False):
        d, F, N = self._frames32(x)
        w = self._win32(window)
        o, st = self.empty((F, n_coeffs), np.float32), self.empty(F, np.int32)
        self._check((self.L.vbx_lpc_burg_f32_wide if wide else self.L.vbx_lpc_burg_f32)(self.ctx, d.ptr, F, N, N, _ptr(w), n_coeffs, o.ptr, st.ptr))
        res = (o.numpy(), st.numpy())
        for b in (d, w, o, st):
            if b is not None:
                b.free()
        return res

    def mfcc_f32(self, x, num_coeffs, freq_bounds, sample_rate, window=None):
        d, F, N = self._frames32(x)
        w = self._win32(window)
        o, st = self.empty((F, num_coeffs), np.float32), self.empty(F, np.int32)
        self._check(self.L.vbx_mfcc_f32(self.ctx, d.ptr, F, N, N, _ptr(w), num_coeffs, freq_bounds[0], freq_bounds[1],
                                        sample_rate, o.ptr, st.ptr))
        res = (o.numpy(), st.numpy())
        for b in (d, w, o, st):
            if b is not None:
                b.free()
        return res

    def pitch_f32(self, x, sample_rate, threshold, fmin, fmax, kmax=8, window=None, wide=False):
        d, F, N = self._frames32(x)
        w = self._win32(window)
        cand, cnt, st = self.empty((F, kmax, 2), np.float32), self.empty(F, np.int32), self.empty(F, np.int32)
        self._check((self.L.vbx_pitch_f32_wide if wide else self.L.vbx_pitch_f32)(self.ctx, d.ptr, F, N, N, _ptr(w), sample_rate, threshold, fmin, fmax, kmax,
                                         cand.ptr, cnt.ptr, st.ptr))
        res = (cand.numpy(), cnt.numpy(), st.numpy())
        for b in (d, w, cand, cnt, st):
            if b is not None:
                b.free()
        return res

    # -- the fused frame loop ---------------------------------------------------------
    def analyze_frames(self, x, params, seg_start=None, frame_len=None, stride=None, n_frames=None, out=None,
                       record_ld=None, status=None):
        """vbx_analyze_frames_f64: pitch + LPC + find_formants + MFCC records [F, record_ld] (see AnalysisParams.columns)."""
        ptr, F, N, S, tmp = self._frames(x, frame_len, stride, n_frames)
        rec = int(self.L.vbx_record_doubles(C.byref(params)))
        ld = record_ld if record_ld is not None else rec + (rec & 1)
        seg = None if seg_start is None else np.ascontiguousarray(seg_start, dtype=np.int64)
        o = out if out is not None else self.empty((F, ld))
        st = status if status is not None else (self.empty((3, F), np.int32) if out is None else None)
        self._check(self.L.vbx_analyze_frames_f64(self.ctx, ptr, F, N, S, C.byref(params),
                                                  None if seg is None else seg.ctypes.data, 0 if seg is None else seg.size,
                                                  _ptr(o), ld, _ptr(st)))
        if out is not None:
            return None
        res = (o.numpy(), st.numpy())
        for d in (o, st, tmp):
            if d is not None:
                d.free()
        return res

    def analyze_frames_pcm16(self, pcm, params, seg_start=None, frame_len=None, stride=None, n_frames=None, out=None,
                             record_ld=None, status=None):
        """vbx_analyze_frames_pcm16: the fused frame loop on 16-bit PCM samples (host int16 array or device buffer)."""
        tmp = None
        if isinstance(pcm, np.ndarray):
            assert pcm.ndim == 1 and frame_len and stride
            n_frames = frame_count(pcm.size, frame_len, stride) if n_frames is None else n_frames
            tmp = self.to_device(pcm, np.int16)
            ptr = tmp.ptr
        else:
            assert frame_len and stride and n_frames is not None
            ptr = _ptr(pcm)
        F = int(n_frames)
        rec = int(self.L.vbx_record_doubles(C.byref(params)))
        ld = record_ld if record_ld is not None else rec + (rec & 1)
        seg = None if seg_start is None else np.ascontiguousarray(seg_start, dtype=np.int64)
        o = out if out is not None else self.empty((F, ld))
        st = status if status is not None else (self.empty((3, F), np.int32) if out is None else None)
        self._check(self.L.vbx_analyze_frames_pcm16(self.ctx, ptr, F, int(frame_len), int(stride), C.byref(params),
                                                    None if seg is None else seg.ctypes.data, 0 if seg is None else seg.size,
                                                    _ptr(o), ld, _ptr(st)))
        if out is not None:
            return None
        res = (o.numpy(), st.numpy())
        for d in (o, st, tmp):
            if d is not None:
                d.free()
        return res

    # -- spectrum.rs: MFCC ------------------------------------------------------------
    def mfcc(self, x, num_coeffs, freq_bounds, sample_rate, frame_len=None, stride=None, n_frames=None,
             window=None, out=None):
        ptr, F, N, S, tmp = self._frames(x, frame_len, stride, n_frames)
        if out is None:
            o, st = self.empty((F, num_coeffs)), self.empty(F, np.int32)
        else:
            o, st = out
        self._check(self.L.vbx_mfcc_f64(self.ctx, ptr, F, N, S, _ptr(window), num_coeffs, freq_bounds[0],
                                        freq_bounds[1], sample_rate, _ptr(o), _ptr(st)))
        if out is not None:
            return None
        res = (o.numpy(), st.numpy())
        for d in (o, st, tmp):
            if d is not None:
                d.free()
        return res

    def dct(self, rows):
        r = np.ascontiguousarray(rows, dtype=np.float64)
        d = self.to_device(r)
        o = self.empty(r.shape)
        self._check(self.L.vbx_dct_f64(self.ctx, d.ptr, r.shape[0], r.shape[1], o.ptr))
        res = o.numpy()
        d.free(); o.free()
        return res

    # -- front end (waves.rs, WAV ingestion) ----------------------------------------------------
    def pcm16_to_f64(self, pcm, out=None):
        """int16 PCM (host array or device buffer + n via `out`) -> f64 / 32767 on the device."""
        tmp = None
        if isinstance(pcm, np.ndarray):
            tmp = self.to_device(pcm, np.int16)
            n, ptr = pcm.size, tmp.ptr
        else:
            n, ptr = out.shape[0], _ptr(pcm)
        o = out if out is not None else self.empty(n)
        self._check(self.L.vbx_pcm16_to_f64(self.ctx, ptr, n, _ptr(o)))
        if tmp is not None:
            self.sync()
            tmp.free()
        return o

    def resample_linear(self, x, ratio, frame_len=None, stride=None, n_frames=None, out=None):
        """find_formants' resample front end (src/lib.rs:57-61): dense [F, ceil(ratio*N)] batch."""
        ptr, F, N, S, tmp = self._frames(x, frame_len, stride, n_frames)
        m = int(self.L.vbx_resampled_len(N, ratio))
        o = out if out is not None else self.empty((F, m))
        self._check(self.L.vbx_resample_linear_f64(self.ctx, ptr, F, N, S, ratio, _ptr(o)))
        return self._finish(o, out, tmp)

    def rms(self, x, frame_len=None, stride=None, n_frames=None, window=None):
        ptr, F, N, S, tmp = self._frames(x, frame_len, stride, n_frames)
        o = self.empty(F)
        self._check(self.L.vbx_rms_f64(self.ctx, ptr, F, N, S, _ptr(window), o.ptr))
        return self._finish(o, None, tmp)

    def preemphasis(self, x, factor, frame_len=None, stride=None, n_frames=None, out=None):
        ptr, F, N, S, tmp = self._frames(x, frame_len, stride, n_frames)
        o = out if out is not None else self.empty((F, N))
        self._check(self.L.vbx_preemphasis_f64(self.ctx, ptr, F, N, S, factor, _ptr(o)))
        return self._finish(o, out, tmp)

    # -- utilities ----------------------------------------------------------------------
    def synth_speech(self, n_samples, sample_offset=0, sample_rate=48000.0, seed=0x5EED0001, out=None):
        o = out if out is not None else self.empty(n_samples)
        self._check(self.L.vbx_synth_speech_f64(self.ctx, _ptr(o), n_samples, sample_offset, sample_rate, seed))
        return o

    def last_unsure_count(self):
        """Frames of the last FFT-path pitch / analyze call that were redone by the direct-sum kernel (test probe)."""
        n = C.c_int32(0)
        self._check(self.L.vbx_internal_last_unsure_count(self.ctx, C.byref(n)))
        return int(n.value)

    def last_burg_direct_count(self):
        """Frames of the last Burg / find_formants call that the one-pass form's guard sent through the direct recursion
        (test probe); -1 if that call did not take the one-pass form."""
        n = C.c_int32(0)
        self._check(self.L.vbx_internal_last_burg_direct_count(self.ctx, C.byref(n)))
        return int(n.value)

    def last_lpc_exact_count(self):
        """Frames of the last fused analyze call whose Levinson row the conditioning probe handed to the double-double recursion
        (k_lpc_exact.hip); -1 if the call wrote no LPC rows or VBX_LPC_EXACT=0."""
        n = C.c_int32(0)
        self._check(self.L.vbx_internal_last_lpc_exact_count(self.ctx, C.byref(n)))
        return int(n.value)

    def last_roots_direct_count(self):
        """Frames of the last find_formants call whose resonances came from the reference's root iteration because the
        conjugate-pair kernel handed them on (test probe); -1 if that call did not use it."""
        n = C.c_int32(0)
        self._check(self.L.vbx_internal_last_roots_direct_count(self.ctx, C.byref(n)))
        return int(n.value)

    def selftest_lanes(self):
        out = np.zeros((2, 64, 8), dtype=np.float64)
        self._check(self.L.vbx_selftest_lanes(self.ctx, out.ctypes.data))
        return out

    def _finish(self, o, out, tmp):
        if out is not None:
            return None
        r = o.numpy()
        o.free()
        if tmp is not None:
            tmp.free()
        return r
