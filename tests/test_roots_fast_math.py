"""The conjugate-pair root finder of k_roots_fast.hip as MATH, on the CPU: its numpy model (tests/roots_fast_model.py)
against the oracle's walk of the reference's own iteration (oracle/vbx_oracle.c: vbxo_find_formants via vbxo_soak), through
the only thing find_formants lets out -- the resonance rows, sorted by frequency (src/lib.rs:94-110)."""
import importlib

import numpy as np

from roots_fast_model import resonance_rows

P = 12


def test_resonance_rows_equal_the_references_on_speech(pkg, oracle):
    synth = importlib.import_module(pkg.__name__ + ".synth")
    for N, H, F in ((512, 512, 4000), (1200, 480, 4000)):
        audio = synth.synth_speech((F - 1) * H + N, 9 * 48000)
        s = oracle.soak(audio, N, H, 0, F, P, 48000.0, oracle.SOAK_FORMANTS)
        rows, count, status, flagged = resonance_rows(s["burg"])
        ok = (s["ff_status"] == 0) & ~flagged
        assert flagged.mean() < 1e-3, flagged.sum()                      # none observed; such a frame is done again by the reference's iteration
        assert np.array_equal(status[ok], s["ff_status"][ok]) and np.array_equal(count[ok], s["res_count"][ok])
        e = s["res"][ok]; g = rows[ok]
        nz = e != 0
        assert np.all(g[~nz] == 0)
        dev = np.abs(g[nz] - e[nz]) / np.abs(e[nz])
        assert dev.max() < 1e-9, dev.max()                               # observed ~1e-11; the gate is 1e-4


def test_degenerate_polynomials(oracle):
    """A zero constant term is the reference's out-of-bounds panic (status 4); non-finite coefficients and polynomials the
    iteration cannot finish are flagged for the reference's own iteration, never answered."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal((8, P)) * 0.3
    a[1, P - 1] = 0.0                                                    # c[0] = a[P-1]
    a[2, 3] = np.nan
    a[3, 0] = np.inf
    rows, count, status, flagged = resonance_rows(a)
    assert status[1] == 4 and count[1] == 0 and not flagged[1]
    assert flagged[2] and flagged[3]
    for f in (0, 4, 5, 6, 7):                                            # ordinary random polynomials: as the oracle
        cl = np.concatenate([a[f][::-1], [1.0]]).astype(np.complex128)
        st, roots = oracle.find_roots_mut(cl)
        assert st == 0 and not flagged[f]
        exp = oracle.to_resonance(roots[roots.imag > 0], 48000.0)
        assert count[f] == exp.shape[0]
        if exp.size:
            assert np.max(np.abs(rows[f, :count[f]] - exp) / np.abs(exp)) < 1e-9
