"""The certificate of the matrix-core candidate pass (myscaledb_amd/csrc/mfma_scan_kernels.hpp) rests on a rounding-error
bound: |approximate distance - canonical distance| <= eps(|x|, |q|, d).  This CPU test replays the approximate
arithmetic in numpy -- split bf16 operands (hi = bf16(v), lo = bf16(v - hi)), the three products xh*qh + xh*ql + xl*qh
accumulated in float32 in several orders, fma-accumulated norms -- against the oracle's canonical distances and checks
that the bound used on the device (same formula, same constants as set_error_model() in msvs_capi.hip and
ivf_rerank_kernel) really is an upper bound, on well-behaved and on nasty inputs.  It cannot see what the MFMA unit does
internally (the bound budgets 2^-23 per accumulated term for that), it pins the algebra and the constants."""
import numpy as np
import pytest

from oracle import oracle as o


def bf16_rne(v):
    """float32 -> nearest bf16 (ties to even), returned as float32."""
    u = np.ascontiguousarray(v, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split(v):
    hi = bf16_rne(v)
    lo = bf16_rne((v - hi).astype(np.float32))  # v - hi is exact in float32
    return hi, lo


def constants(d, scale=1.05):
    c_dot = scale * (3.1 * 2.0 ** -16 + 3.05 * d * 2.0 ** -23)
    c_norm = scale * (d + 8.0) * 2.0 ** -24
    c_canon = scale * 32.0 * 2.0 ** -24
    return c_dot, c_norm, c_canon


def f32_sum(terms, order):
    acc = np.float32(0)
    for t in terms[order]:
        acc = np.float32(acc + t)
    return acc


def approx_dot(x, q, rng):
    """The candidate pass's inner product: all three partial products are exact in float32, the accumulation is not;
    the worst of a few accumulation orders stands in for whatever order the matrix core uses."""
    xh, xl = split(x)
    qh, ql = split(q)
    terms = np.concatenate([(xl * qh), (xh * ql), (xh * qh)]).astype(np.float32)
    exact_terms = np.concatenate([xl.astype(np.float64) * qh, xh.astype(np.float64) * ql, xh.astype(np.float64) * qh])
    assert (terms.astype(np.float64) == exact_terms).all()  # bf16 x bf16 fits float32
    n = len(terms)
    orders = [np.arange(n), np.arange(n)[::-1], rng.permutation(n), np.argsort(-np.abs(terms)), np.argsort(np.abs(terms))]
    return [f32_sum(terms, od) for od in orders]


def fma_norm(v):
    acc = np.float64(0)  # a float32 fma chain rounds once per step: emulate with float64 product + float32 rounding
    a32 = np.float32(0)
    for e in v:
        a32 = np.float32(np.float64(e) * np.float64(e) + np.float64(a32))
    return a32


CASES = [
    ("gaussian", lambda rng, d: (rng.standard_normal(d), rng.standard_normal(d))),
    ("near duplicate", lambda rng, d: (lambda x: (x, x + 1e-4 * rng.standard_normal(d)))(rng.standard_normal(d))),
    ("mixed magnitudes", lambda rng, d: (rng.standard_normal(d) * 10.0 ** rng.integers(-3, 4, d),
                                         rng.standard_normal(d) * 10.0 ** rng.integers(-3, 4, d))),
    ("cancelling", lambda rng, d: (lambda x: (x, -x + 1e-3 * rng.standard_normal(d)))(rng.standard_normal(d) * 50)),
    ("bf16 midpoints", lambda rng, d: ((1.0 + 2.0 ** -8) * (2.0 ** rng.integers(-4, 5, d)) * rng.choice([-1, 1], d),
                                       (1.0 + 3 * 2.0 ** -9) * (2.0 ** rng.integers(-4, 5, d)) * rng.choice([-1, 1], d))),
    ("large", lambda rng, d: (rng.standard_normal(d) * 1e6, rng.standard_normal(d) * 1e6)),
]


@pytest.mark.parametrize("d", [3, 100, 768, 1536])
@pytest.mark.parametrize("name,gen", CASES)
def test_bound_covers_the_split_bf16_candidate_arithmetic(d, name, gen):
    rng = np.random.default_rng(d * 7 + len(name))
    c_dot, c_norm, c_canon = constants(d)
    worst = 0.0
    for _ in range(4):
        x, q = gen(rng, d)
        x, q = x.astype(np.float32), q.astype(np.float32)
        nx, nq = np.sqrt(float(np.dot(x.astype(np.float64), x))), np.sqrt(float(np.dot(q.astype(np.float64), q)))
        true_ip = float(np.dot(x.astype(np.float64), q.astype(np.float64)))
        xn, qn = fma_norm(x), fma_norm(q)
        # what the device plugs into the bound: norms inflated by 0.1 %
        sx, sq = np.sqrt(float(xn) * 1.001), np.sqrt(float(qn) * 1.001)
        assert sx >= nx and sq >= nq
        eps_ip = (c_dot + c_canon) * sx * sq + 1e-30
        eps_l2 = 2.0 * c_dot * sx * sq + c_norm * (sx * sx + sq * sq) + (c_canon + 4e-7) * (sx + sq) ** 2 + 1e-30
        can_ip = float(o.ip(q, x))     # canonical arithmetic (what the re-rank returns)
        can_l2 = float(o.l2sqr(q, x))
        for s in approx_dot(x, q, rng):
            assert abs(float(s) - true_ip) <= c_dot * nx * nq  # the c_dot claim itself
            assert abs(float(s) - can_ip) <= eps_ip
            a_l2 = np.float32(np.float32(np.float64(-2.0) * np.float64(s) + np.float64(xn)) + qn)  # fmaf(-2, S, xn) + qn
            assert abs(float(a_l2) - can_l2) <= eps_l2
            worst = max(worst, abs(float(a_l2) - can_l2) / eps_l2, abs(float(s) - can_ip) / eps_ip)
    assert worst <= 1.0


def test_bf16_emulation_is_round_to_nearest_even():
    v = np.array([1.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, 1.0 + 2.0 ** -9, -1.0 - 2.0 ** -8, 3.3895314e38], np.float32)
    r = bf16_rne(v)
    assert r.tolist()[:5] == [1.0, 1.0, 1.0 + 2.0 ** -6, 1.0, -1.0]  # ties go to the even mantissa
    assert np.isinf(r[5]) or r[5] >= v[5]  # the largest finite float32 rounds up: why huge norms disable the pass
