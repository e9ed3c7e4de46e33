"""The fp32 slab test of the library tree's node boxes (render.hip box_test_f, rows from device_scene.cpp) must be CONSERVATIVE:
whenever the reference's fp64 test (R/AABB.h:68-98, 1/d hoisted) passes a box, the fp32 test of the widened fp32 copy passes it
too -- a node's box only prunes, so extra visits change no frame, a lost visit loses a hit.  Emulated here in numpy (fp32
operations one by one, as the strict build issues them) on a few million random boxes and rays, with the cases that matter:
origins on and inside the boxes, directions with components down to 1e-12 and EXACTLY zero (a Lambertian bounce off an
axis-aligned face has one whenever a uniform comes out as exactly 0.5: dozens of times per C5 frame)."""
import numpy as np

REACH = 5000.0
WIDEN = REACH * 2.0 ** -19          # device_scene.cpp


def _f32(x):
    return x.astype(np.float32)


def _down(v):
    x = _f32(v)
    return np.where(x.astype(np.float64) > v, np.nextafter(x, np.float32(-np.inf)), x)


def _up(v):
    x = _f32(v)
    return np.where(x.astype(np.float64) < v, np.nextafter(x, np.float32(np.inf)), x)


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1000, 1000, (n, 3))
    h = 10 ** rng.uniform(-2, 2.3, (n, 3))
    lo, hi = c - h, c + h
    o = np.where(rng.random((n, 1)) < 0.5, c + rng.uniform(-1.5, 1.5, (n, 3)) * h, rng.uniform(-REACH, REACH, (n, 3)))
    o = np.where(rng.random((n, 3)) < 0.02, lo, o)                         # origins exactly in a face plane
    d = rng.normal(size=(n, 3)) * 10 ** rng.uniform(-12, 0, (n, 3))
    d = np.where(rng.random((n, 3)) < 0.15, 0.0, d)                        # exact zeros, also two of them at once
    d[np.all(d == 0, axis=1)] = (0.3, -0.2, 0.1)
    closest = np.where(rng.random(n) < 0.5, np.inf, 10 ** rng.uniform(-3, 4, n))
    return lo, hi, o, d, closest


def _reference(lo, hi, o, d, closest):
    with np.errstate(all="ignore"):
        inv = 1.0 / d
        t0, t1 = (lo - o) * inv, (hi - o) * inv
        tmin = np.fmax(0.001, np.fmax.reduce(np.fmin(t0, t1), axis=1))      # fmin / fmax drop NaNs like the device's
        tmax = np.fmin(closest, np.fmin.reduce(np.fmax(t0, t1), axis=1))
        return tmax > tmin


def _fp32(lo, hi, o, d, closest, fused):
    flo, fhi = _down(lo - WIDEN), _up(hi + WIDEN)
    with np.errstate(all="ignore"):
        iv = (np.float32(1.0) / _f32(d)).astype(np.float32)
        of = _f32(o)
        if fused:   # lo * (1/d) - o * (1/d) with one rounding: the cheaper form that is NOT used
            c = (of * iv).astype(np.float32)
            a0 = (flo.astype(np.float64) * iv.astype(np.float64) - c.astype(np.float64)).astype(np.float32)
            a1 = (fhi.astype(np.float64) * iv.astype(np.float64) - c.astype(np.float64)).astype(np.float32)
        else:
            a0 = ((flo - of).astype(np.float32) * iv).astype(np.float32)
            a1 = ((fhi - of).astype(np.float32) * iv).astype(np.float32)
        cf = (_f32(closest) * np.float32(1.000002)).astype(np.float32)
        tn = np.fmax(np.fmax.reduce(np.fmin(a0, a1), axis=1), np.float32(0.000999))
        tf = np.fmin(np.fmin.reduce(np.fmax(a0, a1), axis=1), cf)
        return tf > tn


def test_fp32_slab_test_never_loses_a_box_the_fp64_test_passes():
    lost = extra = passed = 0
    for seed in range(3):
        case = _cases(1_000_000, seed)
        ref, got = _reference(*case), _fp32(*case, fused=False)
        lost += int(np.sum(ref & ~got))
        extra += int(np.sum(got & ~ref))
        passed += int(np.sum(ref))
    assert passed > 100_000
    assert lost == 0, f"{lost} boxes lost"
    assert extra < passed          # conservative, not vacuous (tiny boxes in a reach of 5000 are what inflates it here)


def test_the_one_fma_form_loses_boxes_when_a_direction_component_is_zero():
    """Why box_test_f subtracts before it multiplies: with 1/d infinite, lo/d - o/d is inf - inf (NaN, dropped) or -inf - inf
    (-inf: the slab is inverted and the box lost)."""
    case = _cases(1_000_000, 7)
    ref, got = _reference(*case), _fp32(*case, fused=True)
    assert np.sum(ref & ~got) > 0
