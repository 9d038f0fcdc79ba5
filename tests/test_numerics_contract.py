"""The numerics contract (include/cssm_numerics.h) pinned on the CPU: Philox known answers,
elementary functions against glibc/numpy, the fixed-point helpers and the systematic grid count."""
import math

import numpy as np
import pytest

from oracle import oracle


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    ulp = np.abs(np.nextafter(b, np.inf) - b)
    return np.abs(a - b) / ulp


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors for philox4x32-10
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert list(oracle.c_philox(ctr, key)) == want


def test_philox4x32_7_known_answers_the_contracts_generator():
    """Contract v5 draws every variate from SEVEN rounds (the smallest count the Random123 authors report as passing
    BigCrush).  Random123 kat_vectors for philox4x32-7 (the same three counter / key pairs), reproduced by an independent
    pure-Python evaluation of the published round function."""
    kat = [([0, 0, 0, 0], [0, 0], [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a])]

    def py_philox(c, k, rounds):
        c = list(c); k0, k1 = k
        for _ in range(rounds):
            p0 = 0xD2511F53 * c[0]; p1 = 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k0, p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k1, p0 & 0xffffffff]
            k0 = (k0 + 0x9E3779B9) & 0xffffffff; k1 = (k1 + 0xBB67AE85) & 0xffffffff
        return c

    for ctr, key, want in kat:
        assert list(oracle.c_philox_contract(ctr, key)) == want == py_philox(ctr, key, 7)


def test_exp_within_one_ulp_of_libm_and_edge_cases():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-700, 700, 200000), rng.uniform(-50, 0, 200000), rng.normal(0, 1, 100000)])
    assert ulp_diff(oracle.c_exp(x), np.exp(x)).max() <= 1.0
    e = oracle.c_exp(np.array([0.0, -708.5, -1e300, 710.0, np.inf, -np.inf, np.nan]))
    assert e[0] == 1.0 and e[1] == 0.0 and e[2] == 0.0 and e[3] == np.inf and e[4] == np.inf and e[5] == 0.0 and np.isnan(e[6])


def test_log_within_one_ulp_of_libm_and_edge_cases():
    rng = np.random.default_rng(2)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 200000)), rng.random(300000) + 2.0**-53, [1.0, 0.5, 2.0**-53]])
    assert ulp_diff(oracle.c_log(x), np.log(x)).max() <= 1.0
    l = oracle.c_log(np.array([1.0, 0.0, -1.0, np.inf, np.nan, 5e-324]))
    assert l[0] == 0.0 and l[1] == -np.inf and np.isnan(l[2]) and l[3] == np.inf and np.isnan(l[4])
    assert abs(l[5] - math.log(5e-324)) < 1e-12


def test_sincos2pi_accuracy_and_exact_points():
    rng = np.random.default_rng(3)
    u = rng.random(400000)
    s, c = oracle.c_sincos2pi(u)
    a = 2 * np.longdouble("3.14159265358979323846264338327950288") * u.astype(np.longdouble)
    assert np.abs(s - np.sin(a).astype(np.float64)).max() < 3.4e-16   # < 2 ulp near 1 (angle rounded once)
    assert np.abs(c - np.cos(a).astype(np.float64)).max() < 3.4e-16
    s, c = oracle.c_sincos2pi(np.array([0.0, 0.25, 0.5, 0.75, 0.125]))
    np.testing.assert_array_equal(np.abs(s[:4]), [0, 1, 0, 1])
    np.testing.assert_array_equal(np.abs(c[:4]), [1, 0, 1, 0])
    assert s[1] == 1 and c[2] == -1 and s[3] == -1
    assert abs(s[4] - math.sqrt(0.5)) < 2e-16 and abs(c[4] - math.sqrt(0.5)) < 2e-16


def test_sincos_u24_table_and_rotation_accuracy():
    """cssm_sincos_u24 (contract v7: the Box-Muller angle has 24 bits): table entry of the high 8 bits rotated by the low 16 --
    absolute error <= 2^-52 for EVERY one of the 2^24 angles, the table's own points exact, sin^2 + cos^2 = 1 to 2^-51."""
    k = np.arange(1 << 24, dtype=np.uint32)
    s, c = oracle.c_sincos_u24(k)
    a = 2 * np.longdouble("3.14159265358979323846264338327950288") * (k.astype(np.longdouble) / np.longdouble(1 << 24))
    assert np.abs(s - np.sin(a).astype(np.float64)).max() <= 2.0 ** -52   # (measured: exactly 2^-52 at the worst angle)
    assert np.abs(c - np.cos(a).astype(np.float64)).max() <= 2.0 ** -52
    assert np.abs(s * s + c * c - 1.0).max() <= 2.0 ** -51
    q = 1 << 22                                   # quarter turns: table entries 0, 64, 128, 192 hold exact values
    assert (s[0], c[0]) == (0.0, 1.0) and (s[q], c[q]) == (1.0, 0.0) and (s[2 * q], c[2 * q]) == (0.0, -1.0) and (s[3 * q], c[3 * q]) == (-1.0, 0.0)
    # every table point is the correctly rounded value (k = h * 2^16: no rotation, cos delta = 1, sin delta = 0)
    h = np.arange(256, dtype=np.uint32) << 16
    ah = 2 * np.longdouble("3.14159265358979323846264338327950288") * (np.arange(256).astype(np.longdouble) / 256)
    assert np.abs(s[h] - np.sin(ah).astype(np.float64)).max() < 1.2e-16 and np.abs(c[h] - np.cos(ah).astype(np.float64)).max() < 1.2e-16


def test_normal_pairs_are_standard_normal_and_keyed_by_counter():
    z = oracle.c_normals(20260101, 0, 3, 0, 0, 400000).ravel()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert abs(np.mean(z**3)) < 2e-2 and abs(np.mean(z**4) - 3) < 5e-2
    # same counter -> same variate; seed, particle id, step, tag and pair all matter
    base = oracle.c_normals(7, 100, 3, 0, 0, 4)
    np.testing.assert_array_equal(base, oracle.c_normals(7, 100, 3, 0, 0, 4))
    np.testing.assert_array_equal(base[2:], oracle.c_normals(7, 102, 3, 0, 0, 2))  # global id, not local index
    for other in (oracle.c_normals(8, 100, 3, 0, 0, 4), oracle.c_normals(7, 100, 4, 0, 0, 4),
                  oracle.c_normals(7, 100, 3, 1, 0, 4), oracle.c_normals(7, 100, 3, 0, 1, 4)):
        assert not np.array_equal(base, other)


def test_log_unit_table_driven_accuracy_and_sign():
    """The division-free log of Box-Muller's argument: <= 2 ulp on (0, 1], exact 0 at 1, never positive."""
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.random(500000) + 2.0**-53, 1.0 - rng.random(100000) * 2.0**-20, np.exp(-rng.random(100000) * 36.0),
                        [1.0, 1.0 - 2.0**-53, 0.75, 0.5, 2.0**-53, 0.7499999999999999]])
    x = np.minimum(x, 1.0)
    got = oracle.c_log_unit(x)
    assert ulp_diff(got, np.log(x)).max() <= 2.0
    assert got.max() <= 0.0 and oracle.c_log_unit(np.array([1.0]))[0] == 0.0


def test_uniform_for_resampling_is_in_unit_interval():
    u = np.array([oracle.lib().oracle_c_u(5, s) for s in range(2000)])
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.03


def test_lgamma_integer():
    for k in list(range(0, 60)) + [255, 256, 1000, 123456]:
        assert abs(oracle.lib().oracle_c_lgamma_kp1(k) - math.lgamma(k + 1.0)) <= 2e-14 * max(1.0, math.lgamma(k + 1.0))


def test_fixed_point_roundtrip_and_truncation():
    f = oracle.lib().oracle_c_fix_roundtrip
    for w in (1.0, 0.5, 0.75, 2.0**-40, 2.0**-96, 1 - 2.0**-53, 0.1):
        got = f(w)
        assert got <= w and w - got < 2.0**-96
    assert f(1.0) == 1.0 and f(2.0**-96) == 2.0**-96
    assert f(2.0**-97) == 0.0 and f(0.0) == 0.0 and f(-1.0) == 0.0 and f(float("nan")) == 0.0 and f(float("inf")) == 0.0


def test_sys_count_matches_brute_force():
    rng = np.random.default_rng(4)
    cnt = oracle.lib().oracle_c_sys_count
    for _ in range(3000):
        n = int(rng.integers(1, 400))
        u = float(rng.random())
        C = float(rng.random() * 1.001)
        brute = sum(1 for i in range(n) if (u + i) / n <= C)
        assert cnt(C, u, n) == brute
    assert cnt(1.0, 0.999, 16) == 16 and cnt(0.0, 0.5, 16) == 0 and cnt(0.0, 0.0, 16) == 1


def test_fixed_point_conversion_is_the_exact_floor():
    """cssm_fix_from_double(w) = floor(w * 2^96) for finite 0 <= w < 2^9, else 0 -- against Python integers."""
    from fractions import Fraction
    rng = np.random.default_rng(12)
    w = np.concatenate([rng.random(2000), rng.random(2000) * 2.0 ** -rng.integers(0, 1100, 2000), rng.random(300) * 511.9,
                        [0.0, 1.0, 2.0 ** -96, 2.0 ** -97, 2.0 ** -1074, 511.99999999999994, 512.0, 600.0, -1.0, np.inf, np.nan]])
    out = oracle.c_fix(w)
    for x, (lo, hi) in zip(w, out):
        want = int(Fraction(x) * 2 ** 96) if (np.isfinite(x) and 0 <= x < 512) else 0
        assert (int(hi) << 64) | int(lo) == want, x


def test_paired_streams_share_blocks_without_reuse():
    """Ordinary steps / initial draw: particles 2m and 2m+1 share stream m; particle gid owns normals (gid&1)*d + k of it."""
    seed, step, tag = 99, 7, 0
    for d in (1, 2, 3, 4, 9):
        z = oracle.c_paired_normals(seed, 10, step, tag, d, 6)          # particles 10..15 = pairs 5, 6, 7
        blocks = oracle.c_normals(seed, 5, step, tag, 0, 3)             # block 0 of streams 5, 6, 7
        for m in range(3):
            flat = np.concatenate([oracle.c_normals(seed, 5 + m, step, tag, b, 1)[0] for b in range(d)])   # blocks 0..d-1 of stream 5+m
            np.testing.assert_array_equal(z[2 * m], flat[:d])
            np.testing.assert_array_equal(z[2 * m + 1], flat[d:2 * d])
        assert blocks.shape == (3, 2)
        assert len(np.unique(z)) == z.size                              # nothing drawn twice
