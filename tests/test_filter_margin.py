"""The sphere-list scan's conservative filter (render.hip scan_ray / filter_s / filter_q / filter_behind), restated in
Python with the kernel's operation order -- fused multiply-adds evaluated exactly (fractions) and rounded once -- against
the reference's discriminant test (R/Sphere.h:28-41) in plain double arithmetic, on rays built to graze spheres.

The property the kernel relies on: whenever the reference's `disc > 0` holds (and the sphere is not "outside and behind"),
the filter passes the sphere and does not call it behind.  Then every sphere the reference could accept reaches the
drain, which runs the reference's arithmetic, and the frame is the exact scan's bit for bit (the GPU tests compare those).
"""
import math
from fractions import Fraction

import numpy as np
import pytest


def fma(a, b, c):
    if not (math.isfinite(a) and math.isfinite(b) and math.isfinite(c)):
        return a * b + c                                    # infinities / NaNs propagate as in hardware
    return float(Fraction(a) * Fraction(b) + Fraction(c))   # one rounding, like v_fma_f64


def dot(a, b):                                              # strict build: no contraction, left to right
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def scan_ray(o, d, reach):
    a = dot(d, d)
    oo = dot(o, o)
    w = math.sqrt(oo) + reach
    inv = 1.0 / math.sqrt(a)
    u = [inv * x for x in d]
    od = dot(o, u)
    p2 = [2.0 * (o[i] - od * u[i]) for i in range(3)]
    root_m = 2.0 ** -20 * w
    nthr = root_m * root_m - (oo - od * od)
    return dict(u=u, p2=p2, od=od, root_m=root_m, nthr=nthr)


def filter_pass(f, c, k):
    s = fma(c[2], f["u"][2], fma(c[1], f["u"][1], c[0] * f["u"][0]))
    q = fma(s, s, fma(f["p2"][2], c[2], fma(f["p2"][1], c[1], fma(f["p2"][0], c[0], f["nthr"]))))
    passed = q > k
    bu = f["od"] - s
    behind = passed and bu > f["root_m"] and fma(bu, bu, k - q) > 0.0
    return passed, behind


def reference(o, d, c, r2):
    oc = [o[i] - c[i] for i in range(3)]
    a = dot(d, d)
    b = dot(oc, d)
    cc = dot(oc, oc) - r2
    disc = b * b - a * cc
    return disc > 0.0, (b > 0.0 and cc > 0.0)


def host_row(c, r):
    r2 = r * r
    k = float(Fraction(c[0]) ** 2 + Fraction(c[1]) ** 2 + Fraction(c[2]) ** 2 - Fraction(r2))
    reach = (math.sqrt(dot(c, c)) + math.sqrt(r2)) * (1.0 + 2.0 ** -40)
    return r2, k, reach


@pytest.mark.parametrize("offset,scale", [((0.0, 0.0, 0.0), 1.0), ((13.0, 2.0, 3.0), 1.0), ((3000.0, -2000.0, 5000.0), 1.0),
                                          ((1.0e6, 2.0e6, -3.0e6), 1.0), ((0.0, -1000.0, 0.0), 1000.0)])
def test_filter_never_rejects_what_the_reference_accepts(offset, scale):
    rng = np.random.default_rng(7)
    grazing = accepted = passed_in_vain = 0
    for trial in range(1500):
        c = [offset[i] + float(rng.uniform(-11, 11)) for i in range(3)]
        r = scale * float(rng.choice([0.2, 1.0, 0.5]))
        r2, k, reach_c = host_row(c, r)
        reach = max(reach_c, 2000.0 * (trial % 2))          # with and without a big ground sphere elsewhere in the scene
        # a ray whose line passes the sphere at distance r * (1 + eps), eps from far outside to far inside through zero
        o = [offset[i] + float(rng.uniform(-15, 15)) for i in range(3)]
        tdir = rng.normal(size=3)
        to_c = np.array(c) - np.array(o)
        perp = np.cross(to_c, tdir)
        perp /= np.linalg.norm(perp)
        eps = float(rng.choice([0.0, 1e-16, -1e-16, 1e-14, -1e-14, 1e-12, -1e-12, 1e-9, -1e-9, 1e-6, -1e-6, 1e-3, -1e-3, 0.3, -0.3]))
        target = np.array(c) + perp * r * (1.0 + eps)
        d = (target - np.array(o)) * float(rng.uniform(0.2, 3.0)) * float(rng.choice([1.0, -1.0]))
        d = [float(x) for x in d]
        f = scan_ray(o, d, reach)
        for kk in (k, np.nextafter(k, np.inf), np.nextafter(k, -np.inf)):   # the host's K may be an ulp off the rounded one
            ok, behind = filter_pass(f, c, float(kk))
            acc, ref_behind = reference(o, d, c, r2)
            if acc and not ref_behind:
                accepted += 1
                assert ok and not behind, (o, d, c, r, eps)
            if behind:
                assert (not acc) or ref_behind, "the filter calls a sphere behind that the reference would test"
            if ok and not acc:
                passed_in_vain += 1
        grazing += abs(eps) <= 1e-9
    assert accepted > 300 and grazing > 500
    # the filter is tight where coordinates are moderate: spheres passed in vain are the grazing ones only
    if max(abs(x) for x in offset) < 100 and scale == 1.0:
        assert passed_in_vain < 0.4 * 3 * 1500


def test_degenerate_rays_pass_everything():
    for d in ([0.0, 0.0, 0.0], [1e-200, 0.0, 0.0], [float("nan"), 1.0, 0.0]):
        a = dot(d, d)
        assert not (a > 1e-280 and a < 1e280)   # scan_ray's `sane` is false: u = p2 = 0, nthr = +inf, root_m = +inf
        f = dict(u=[0.0] * 3, p2=[0.0] * 3, od=0.0, root_m=math.inf, nthr=math.inf)
        ok, behind = filter_pass(f, [1.0, 2.0, 3.0], 13.0)
        assert ok and not behind


# ---- the packed fp32 form (render.hip scan_ray32 / filter_pairs): the same quantities, every operation rounded to fp32 ----
F = np.float32


def fma32(a, b, c):
    """v_pk_fma_f32 / v_fma_f32: one rounding to fp32 (the exact product of two fp32 numbers fits a double; the sum is rounded
    to double and then to fp32 -- a double rounding that can differ from the fused one by half an fp32 ulp in 2^-29 of the cases:
    irrelevant to a margin that is 64 ulps wide)."""
    with np.errstate(all="ignore"):
        return F(np.float64(a) * np.float64(b) + np.float64(c))


def scan_ray32(o, d, reach):
    a = dot(d, d)
    oo = dot(o, o)
    w = math.sqrt(oo) + reach
    inv = 1.0 / math.sqrt(a)
    u = [inv * x for x in d]
    od = dot(o, u)
    p2 = [2.0 * (o[i] - od * u[i]) for i in range(3)]
    root_m = 2.0 ** -9 * w
    nthr = root_m * root_m - (oo - od * od)
    return dict(u=[F(x) for x in u], p2=[F(x) for x in p2], od=F(od), root_m=F(root_m), nthr=F(nthr))


def filter_pass32(f, c, k):
    cf = [F(x) for x in c]
    with np.errstate(all="ignore"):
        s = fma32(cf[2], f["u"][2], fma32(cf[1], f["u"][1], F(cf[0] * f["u"][0])))
        q = fma32(s, s, fma32(f["p2"][2], cf[2], fma32(f["p2"][1], cf[1], fma32(f["p2"][0], cf[0], f["nthr"]))))
        kf = F(k)
        passed = bool(q > kf)
        bu = F(f["od"] - s)
        behind = passed and bool(bu > f["root_m"]) and bool(fma32(bu, bu, F(kf - q)) > 0)
    return passed, behind


@pytest.mark.parametrize("offset,scale,spread", [((0.0, 0.0, 0.0), 1.0, 11.0), ((13.0, 2.0, 3.0), 1.0, 11.0), ((300.0, -200.0, 500.0), 1.0, 11.0),
                                                 ((0.0, 0.0, 0.0), 0.01, 3.0), ((0.0, 0.0, 0.0), 50.0, 400.0)])
def test_packed_fp32_filter_never_rejects_what_the_reference_accepts(offset, scale, spread):
    """Same property, fp32 margins: the spheres this filter decides lie within `reach32` of the origin (the host leaves the
    outliers, k = -inf, to the exact test); rays come from anywhere -- camera, surfaces, far away."""
    rng = np.random.default_rng(11)
    accepted = grazing = passed_in_vain = total = 0
    for trial in range(1500):
        c = [offset[i] + float(rng.uniform(-spread, spread)) for i in range(3)]
        r = scale * float(rng.choice([0.2, 1.0, 0.5]))
        r2, k, reach_c = host_row(c, r)
        reach32 = max(reach_c, (math.sqrt(dot(offset, offset)) + 1.8 * spread + scale) * (1.0 + 2.0 ** -20))   # the bulk's reach
        far = float(rng.choice([1.0, 1.0, 10.0, 1000.0]))                                                      # origins well outside the bulk too
        o = [offset[i] + far * float(rng.uniform(-15, 15)) for i in range(3)]
        tdir = rng.normal(size=3)
        to_c = np.array(c) - np.array(o)
        perp = np.cross(to_c, tdir)
        perp /= np.linalg.norm(perp)
        eps = float(rng.choice([0.0, 1e-16, -1e-16, 1e-12, -1e-12, 1e-9, -1e-9, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-3, -1e-3, 0.3, -0.3]))
        target = np.array(c) + perp * r * (1.0 + eps)
        d = (target - np.array(o)) * float(rng.uniform(0.2, 3.0)) * float(rng.choice([1.0, -1.0]))
        d = [float(x) for x in d]
        f = scan_ray32(o, d, reach32)
        ok, behind = filter_pass32(f, c, k)
        acc, ref_behind = reference(o, d, c, r2)
        total += 1
        if acc and not ref_behind:
            accepted += 1
            assert ok and not behind, (o, d, c, r, eps)
        if behind:
            assert (not acc) or ref_behind, "the fp32 filter calls a sphere behind that the reference would test"
        if ok and not acc:
            passed_in_vain += 1
        grazing += abs(eps) <= 1e-6
    assert accepted > 300 and grazing > 500


def test_packed_fp32_filter_is_sharp_on_the_benchmark_scene():
    """C2's geometry: small spheres within 12 of the origin, camera at (13, 2, 3): the filter passes, of the spheres a ray misses,
    only those it misses by a few per cent of their radius (M / 2r with M = 2^-18 W^2, W ~ 30)."""
    rng = np.random.default_rng(5)
    vain = misses = 0
    for trial in range(4000):
        c = [float(rng.uniform(-11, 11)), 0.2, float(rng.uniform(-11, 11))]
        r2, k, reach_c = host_row(c, 0.2)
        o = [13.0, 2.0, 3.0] if trial % 2 else [float(rng.uniform(-11, 11)), float(rng.uniform(0, 2)), float(rng.uniform(-11, 11))]
        d = [float(x) for x in rng.normal(size=3)]
        f = scan_ray32(o, d, 16.5)
        ok, behind = filter_pass32(f, c, k)
        acc, _ = reference(o, d, c, r2)
        if not acc:
            misses += 1
            vain += ok
        else:
            assert ok
    assert misses > 3000 and vain < 0.01 * misses
