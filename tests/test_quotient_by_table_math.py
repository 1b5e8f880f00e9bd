"""The lag-window divide of the fused kernels (vbx_spectral.hpp quotient_by_table, round 6) as MATH, on the CPU: for a divisor b that
comes from a table with its correctly rounded reciprocal y = RN(1 / b) beside it,
    q0 = RN(a y),  rem = RN(a - q0 b)  (one FMA),  q = RN(q0 + rem y)  (one FMA)
is the IEEE quotient RN(a / b).  Checked in exact rational arithmetic (fractions.Fraction; float(Fraction) rounds to nearest even) on the
divisors the kernels really use -- the lag windows of src/periodic.rs:232-252 at the bench's frame lengths -- and numerators spread over
the lag curve's range.  (The GPU side of the claim: tools/experiments/bitcompare_libs.py, every output of 1.2 M frames bit for bit the
IEEE-division build's, profiles/r06_headline/r06c_bit_identical_to_round5.txt.)"""
from fractions import Fraction

import numpy as np


def _fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def test_quotient_by_table_is_the_ieee_quotient_on_the_lag_windows(oracle):
    rng = np.random.default_rng(6)
    checked = 0
    for n in (1200, 1103, 512, 2048):
        lw = oracle.window("hanning_lag", n)
        rw = 1.0 / lw                                                      # the host's IEEE division: correctly rounded reciprocals
        assert np.all(np.isfinite(rw))
        for i in rng.choice(n, size=160, replace=False):
            b, y = float(lw[i]), float(rw[i])
            assert y == float(1 / Fraction(b))                             # (the table's premise)
            for a in np.concatenate([rng.standard_normal(6) * 10.0 ** rng.integers(-9, 1, 6), [0.0, 1.0, -1.0]]):
                a = float(a)
                q0 = a * y
                rem = _fma(-q0, b, a)
                q = _fma(rem, y, q0)
                assert q == float(Fraction(a) / Fraction(b)), (n, int(i), a, b)
                checked += 1
    assert checked > 5000
