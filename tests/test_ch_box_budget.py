"""CPU check of the error budget behind the centre / half-extent culling boxes of the MIXED program (DESIGN.md 4a,
rtk_api.cpp build_mixed_program, rtk_trace.hip slab_test32_ch / slab_test32_chs / rescale32).

The float test only decides which primitives get tested, so it must be CONSERVATIVE: whenever aabb::hit (aabb.h:61-85) in
double passes a ray through the reference's box within (tmin, tmax), the float test on the grown record must pass it too.
This file restates both sides in numpy -- the record builder's formula and the kernel's float arithmetic, with v_rcp_f32
modelled as the correctly rounded reciprocal pushed one ulp either way (the instruction is good to one ulp) -- and hammers
the pair with rays that graze box faces, edges and corners at coordinates up to the scene extent.  It pins the arithmetic of
the budget, not the device code (the GPU parity tests compare whole images and work counters bit for bit)."""
import numpy as np
import pytest

F = np.float32


def fma32(a, b, c):
    """float fma of float operands: the product of two floats is exact in double, the sum is rounded once to double
    (2^-53: far below anything checked here) and once to float."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def round_up32(x):
    f = x.astype(F)
    low = f.astype(np.float64) < x
    return np.where(low, np.nextafter(f, F(np.inf)), f).astype(F)


def build_record(lo, hi, extent):
    """rtk_api.cpp build_mixed_program, RTK_CH_BOX: float centre, half-extent from that centre grown by
    2^-21 (|c| + h) + 2^-20 extent, rounded up."""
    c = 0.5 * (lo + hi)
    cf = c.astype(F)
    h = np.maximum(hi - cf.astype(np.float64), cf.astype(np.float64) - lo)
    grown = h + np.ldexp(np.abs(cf.astype(np.float64)) + h, -21) + np.ldexp(extent, -20) + 1e-300
    return cf, round_up32(grown * (1.0 + 1e-7))


def exact_slab(lo, hi, o, d, tmin, tmax):
    """aabb::hit for rays with finite non-zero 1/d on every axis (the only ones the float loop takes)."""
    inv = 1.0 / d
    t0, t1 = (lo - o) * inv, (hi - o) * inv
    near = np.maximum(np.minimum(t0, t1).max(axis=1), tmin)
    far = np.minimum(np.maximum(t0, t1).min(axis=1), tmax)
    return far > near  # aabb.h: "if (ray_t.max <= ray_t.min) return false"


def float_slab(cf, hf, o, d, tmax, extent, ulp_push, scaled):
    d32 = d.astype(F)
    inv = (F(1) / d32).astype(F)
    inv = np.where(ulp_push > 0, np.nextafter(inv, F(np.inf)), np.where(ulp_push < 0, np.nextafter(inv, F(-np.inf)), inv)).astype(F)
    oi = (o.astype(F) * inv).astype(F)
    f = F(tmax)
    tmax32 = (f + np.abs(f) * F(2.3841858e-07)).astype(F) if np.isfinite(tmax) else F(np.inf)  # above()
    tmin32 = F(0.001) - F(0.001) * F(2.3841858e-07)                                              # below()
    if scaled:  # rescale32
        bound = (F(4) * F(extent) * np.abs(inv[:, 0])).astype(F)
        end = np.minimum(bound, tmax32).astype(F)
        s = ((F(1) / end).astype(F) * F(0.99999976)).astype(F)
        s = np.nextafter(s, F(np.inf))  # v_rcp_f32 one ulp HIGH: the adverse direction
        inv = (inv * s[:, None]).astype(F)
        oi = (oi * s[:, None]).astype(F)
    tc = fma32(cf, inv, -oi)
    near = fma32(-hf, np.abs(inv), tc).max(axis=1)
    far = fma32(hf, np.abs(inv), tc).min(axis=1)
    if scaled:
        return np.clip(far, 0, 1) > np.clip(near, 0, 1)
    return np.minimum(far, tmax32) >= np.maximum(near, tmin32)


def grazing_cases(rng, n, extent):
    """Boxes from sphere-sized to scene-sized anywhere in the scene, rays aimed at points ON their surface (faces, edges,
    corners) from origins anywhere within the extent -- the cases in which a rounding decides."""
    size = 10.0 ** rng.uniform(-2.5, np.log10(extent), (n, 1))
    c = rng.uniform(-1, 1, (n, 3)) * np.maximum(extent - size, 0.0)
    half = size * rng.uniform(0.05, 1.0, (n, 3)) * 0.5
    lo, hi = c - half, c + half
    u = rng.uniform(0, 1, (n, 3))
    snap = rng.integers(0, 3, (n, 3))  # per axis: 0 = inside, 1 = on the low face, 2 = on the high face
    target = np.where(snap == 1, lo, np.where(snap == 2, hi, lo + u * (hi - lo)))
    o = rng.uniform(-extent, extent, (n, 3))
    near_box = rng.uniform(0, 1, (n, 1)) < 0.5  # half of them start close by (secondary rays do)
    o = np.where(near_box, np.clip(c + rng.normal(0, 3, (n, 3)) * size, -extent, extent), o)
    d = (target - o) * 10.0 ** rng.uniform(-1, 1, (n, 1))
    keep = (np.abs(d) > 1e-9).all(axis=1)
    return lo[keep], hi[keep], o[keep], d[keep]


@pytest.mark.parametrize("scaled", [False, True])
@pytest.mark.parametrize("extent", [1001.0, 30.0])
def test_float_box_test_never_rejects_what_the_reference_accepts(scaled, extent):
    rng = np.random.default_rng(20261005)
    lo, hi, o, d = grazing_cases(rng, 400_000, extent)
    cf, hf = build_record(lo, hi, extent)
    checked = 0
    for tmax in (np.inf, None):  # no hit yet; a closest hit at a random distance around the box
        if tmax is None:
            inv = 1.0 / d
            entry = np.minimum((lo - o) * inv, (hi - o) * inv).max(axis=1)
            tm = np.abs(entry) * 10.0 ** rng.uniform(-0.3, 0.3, len(entry)) + 0.002
        for push in (-1, 0, 1):
            for k in range(0, len(lo), 100_000):
                sl = slice(k, k + 100_000)
                tmx = np.full(len(lo[sl]), np.inf) if tmax is not None else tm[sl]
                want = np.zeros(len(tmx), bool)
                got = np.zeros(len(tmx), bool)
                for t in np.unique(tmx) if tmax is not None else [None]:
                    if t is None:  # per-ray tmax: evaluate row by row in groups of equal value (vectorised through a loop over 64 quantiles)
                        q = np.quantile(tmx, np.linspace(0, 1, 33))
                        for a, b in zip(q[:-1], q[1:]):
                            m = (tmx >= a) & (tmx <= b)
                            if not m.any():
                                continue
                            tcap = float(b)  # a common, larger end for the group: both tests see the same interval
                            want[m] = exact_slab(lo[sl][m], hi[sl][m], o[sl][m], d[sl][m], 0.001, tcap)
                            got[m] = float_slab(cf[sl][m], hf[sl][m], o[sl][m], d[sl][m], tcap, extent, push, scaled)
                    else:
                        want = exact_slab(lo[sl], hi[sl], o[sl], d[sl], 0.001, np.inf)
                        got = float_slab(cf[sl], hf[sl], o[sl], d[sl], np.inf, extent, push, scaled)
                missed = want & ~got
                assert not missed.any(), (int(missed.sum()), lo[sl][missed][0], hi[sl][missed][0], o[sl][missed][0], d[sl][missed][0])
                checked += int(want.sum())
    assert checked > 500_000  # the cases really are hits of the exact test, most of them by a hair


def test_the_margin_is_needed():
    """Negative control: the same records WITHOUT the growth lose hits the reference accepts -- the test above is not vacuous."""
    rng = np.random.default_rng(7)
    extent = 1001.0
    lo, hi, o, d = grazing_cases(rng, 200_000, extent)
    c = 0.5 * (lo + hi)
    cf = c.astype(F)
    hf = (np.maximum(hi - cf.astype(np.float64), cf.astype(np.float64) - lo)).astype(F)  # rounded to nearest, not grown
    want = exact_slab(lo, hi, o, d, 0.001, np.inf)
    got = float_slab(cf, hf, o, d, np.inf, extent, -1, True)
    assert (want & ~got).sum() > 10
