"""Error behaviour of the C ABI on a live context: API misuse returns VBX_E_INVALID with a message
(never a crash), shapes outside the documented limits are refused, empty batches are no-ops."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rc(vb, fn, *args):
    rc = fn(vb.ctx, *args)
    msg = vb.L.vbx_last_error(vb.ctx)
    return rc, (msg.decode() if msg else "")


def test_misuse_is_reported_not_fatal(vb, pkg):
    L = vb.L
    x = vb.zeros(4096)
    out = vb.zeros(4096)
    rc, msg = _rc(vb, L.vbx_autocorrelate_f64, x.ptr, 4, 512, 512, None, 0, out.ptr)          # n_lags = 0
    assert rc == -1 and "n_lags" in msg
    rc, msg = _rc(vb, L.vbx_autocorrelate_f64, x.ptr, 4, 512, 512, None, 600, out.ptr)        # n_lags > frame_len
    assert rc == -1
    rc, msg = _rc(vb, L.vbx_autocorrelate_f64, None, 4, 512, 512, None, 13, out.ptr)          # null frames
    assert rc == -1 and "null" in msg
    # frame_len: every f64 entry point takes up to VBX_MAX_LONG_FRAME_LEN samples (beyond 4096: the tiled kernels of k_long.hip),
    # the f32 instantiation VBX_MAX_FRAME_LEN
    rc, msg = _rc(vb, L.vbx_autocorrelate_f64, x.ptr, 1, (1 << 26) + 1, 1, None, 13, out.ptr)
    assert rc == -1 and "frame_len" in msg
    rc, msg = _rc(vb, L.vbx_pitch_f64, x.ptr, 1, (1 << 26) + 1, 1, None, 48000.0, 0.2, 75.0, 600.0, 1, out.ptr, None, None)
    assert rc == -1 and "frame_len" in msg
    rc, msg = _rc(vb, L.vbx_autocorrelate_f32, x.ptr, 1, 5000, 5000, None, 13, out.ptr)
    assert rc == -1 and "frame_len" in msg
    rc, msg = _rc(vb, L.vbx_pitch_f64, x.ptr, 1, 5000, 5000, None, 48000.0, 0.2, 75.0, 600.0, 2000, out.ptr, None, None)   # kmax > 5000 / 4 + 2
    assert rc == -1 and "kmax" in msg
    rc, msg = _rc(vb, L.vbx_lpc_burg_f64, x.ptr, 4, 512, 512, None, 63, out.ptr, None)       # order > 62
    assert rc == -1
    rc, msg = _rc(vb, L.vbx_pitch_f64, x.ptr, 4, 512, 512, None, 48000.0, 0.2, 75.0, 600.0, 1027, out.ptr, None, None)
    assert rc == -1 and "kmax" in msg
    est = np.array([[320.0, 1.0]] * 7)
    rc, msg = _rc(vb, L.vbx_find_formants_f64, x.ptr, 4, 512, 512, 48000.0, 12, None, 0, est.ctypes.data, 7,
                  out.ptr, None, None, None, None)
    assert rc == -1 and "n_est" in msg
    seg = np.array([1, 2], dtype=np.int64)                                                   # seg_start[0] != 0
    rc, msg = _rc(vb, L.vbx_find_formants_f64, x.ptr, 4, 512, 512, 48000.0, 12, seg.ctypes.data, 2,
                  est.ctypes.data, 4, out.ptr, None, None, None, None)
    assert rc == -1 and "seg_start" in msg
    rc, msg = _rc(vb, L.vbx_preemphasis_f64, x.ptr, 4, 512, 480, 0.1, x.ptr)                  # in place on a strided view
    assert rc == -1
    # the context is still usable
    assert vb.autocorrelate(np.ones((2, 64)), 3).shape == (2, 3)
    x.free(); out.free()


def test_empty_batches(vb):
    L = vb.L
    assert L.vbx_lpc_burg_f64(vb.ctx, None, 0, 512, 512, None, 12, None, None) == 0
    assert L.vbx_mfcc_f64(vb.ctx, None, 0, 512, 512, None, 13, 100.0, 8000.0, 48000.0, None, None) == 0
    assert L.vbx_find_roots_c64(vb.ctx, None, 0, 13, None) == 0
    assert L.vbx_pcm16_to_f64(vb.ctx, None, 0, None) == 0
    assert vb.find_formants(np.zeros((0, 512)), 48000.0, 12, np.array([[320.0, 1.0]] * 4))["formants"].shape == (0, 4, 2)


def test_window_table_rejects_unknown_kind(pkg):
    out = np.empty(8)
    assert pkg.load_library().vbx_window_table_f64(17, 8, out.ctypes.data) == -1
