"""HIP megakernel against the CPU oracle on scenes built constructor by constructor on BOTH sides.

The built-in scenes (ids 0..11) are transcribed twice, once per side; the scenes here are written once, as a
function of a scene object, and run against raytracinginoneweekendincuda_amd.Scene (C-ABI, include/rtow.h) and
against conftest.OracleScene (oracle_c_* constructors of oracle/rtow_oracle.c).  They cover what no built-in scene
pins: the duplicated span-1 BVH leaf holding a ConstantMedium (SURVEY Q7, R/BvhNode.h:63-67 + R/ConstantMedium.h:52-94),
and quads in every axis pairing / boxes thin, far and instanced.
"""
import numpy as np
import pytest

from conftest import build_both

pytestmark = pytest.mark.gpu

TOL = 1e-5
W, H, SPP = 64, 32, 4


def compare(got, want):
    diff = np.abs(got - want)
    exact = np.mean(np.all(got.view(np.uint64) == want.view(np.uint64), axis=-1))
    within = np.mean(np.all(diff <= TOL, axis=-1))
    return exact, within, diff.max()


def check(build, w=W, h=H, spp=SPP, min_exact=0.99, variants=(0, 1), min_within_fast=0.995):
    prod, orc = build_both(build)
    want, stats = orc.render(w, h, spp, want_stats=True)
    for variant in variants:
        got, st = prod.render(w, h, spp, variant=variant)
        exact, within, worst = compare(got, want)
        print(f"variant {variant}: kernel kind {st.kernel_kind}, bit-exact {exact:.4f}, within {within:.4f}, max |d| {worst:.3g}")
        assert np.isfinite(got).all()
        if variant == 0:
            assert st.rays == stats["rays"], "ray counter differs from the oracle's RayColor iterations"
            assert within >= 0.999 and exact >= min_exact
        else:
            assert within >= min_within_fast
    return want


# ---- Q7: the duplicated leaf of a span-1 BvhNode ----
def _one_medium(world):
    def build(s, Rng):
        ball = s.Sphere((0, 0, -3), 1.0, s.Dielectric(1.5))
        fog = s.ConstantMedium(ball, 0.8, (0.9, 0.2, 0.2))
        items = [fog]
        s.SetWorld(s.BvhNode(items) if world == "bvh" else s.HittableList(items))
        s.Camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 60, W / H, 0.0, 10.0)
        s.Commit()
    return build


def test_span1_bvh_medium_leaf_is_hit_twice_like_the_oracle():
    """BvhNode over one ConstantMedium: left == right == the leaf, so the medium is queried twice per visit and draws
    twice (the second call with tMax = the first hit).  The oracle walks the pointer tree exactly like
    R/BvhNode.h:109-155; the kernel's threaded walk must consume the same draws."""
    bvh = check(_one_medium("bvh"), spp=8)
    lst = check(_one_medium("list"), spp=8)
    assert not np.array_equal(bvh, lst), "the duplicated leaf must show (the list world hits the medium once)"


def test_span1_medium_leaf_inside_a_larger_bvh():
    """Three leaves: the median split leaves one span-1 node (the box medium, sorted first on x) beside a span-2 node."""
    def build(s, Rng):
        white = s.Lambertian((0.73, 0.73, 0.73))
        smoke = s.ConstantMedium(s.MakeBox((-3.2, -1, -4), (-1.6, 0.8, -2.5), white), 0.9, (0.1, 0.1, 0.1))
        items = [s.Sphere((0.3, 0, -3), 0.8, s.Metal((0.8, 0.7, 0.6), 0.2)),
                 smoke,
                 s.Sphere((2.4, 0, -3.5), 0.9, s.Lambertian((0.2, 0.4, 0.8)))]
        s.SetWorld(s.BvhNode(items))
        s.Camera((0, 0.5, 2), (0, 0, -3), (0, 1, 0), 60, W / H, 0.0, 10.0)
        s.Commit()
    check(build, spp=8)


# ---- quads and boxes ----
def _quad_zoo(world_kind):
    def build(s, Rng):
        red, green, grey = s.Lambertian((0.65, 0.05, 0.05)), s.Lambertian((0.12, 0.45, 0.15)), s.Lambertian((0.73, 0.73, 0.73))
        metal, glass, light = s.Metal((0.8, 0.8, 0.9), 0.1), s.Dielectric(1.5), s.DiffuseLight((4.0, 4.0, 4.0))
        items = []
        e = [(1.5, 0, 0), (0, 1.5, 0), (0, 0, 1.5)]
        k = 0
        for a in range(3):
            for p in range(3):
                if p == a:
                    continue
                q = 3 - a - p
                for su, sv in ((1, 1), (-1, 1), (1, -1)):
                    u = tuple(su * c for c in e[p])
                    v = tuple(sv * c for c in e[q])
                    org = [-6.0 + 1.7 * (k % 7), -2.0 + 1.9 * (k // 7), -3.0 - 0.3 * k]
                    items.append(s.Quad(org, u, v, (red, green, grey, metal)[k % 4]))
                    k += 1
        items.append(s.Quad((-1.0, 3.5, -4.0), (2.0, 0.3, 0.1), (0.2, 1.5, -0.4), green))          # not axis-aligned
        items.append(s.Quad((-8.0, 6.0, -12.0), (16.0, 0, 0), (0, 0, 14.0), light))
        items.append(s.MakeBox((-5.0, -3.0, -2.0), (-3.5, -1.0, -0.5), grey))                      # plain box leaf
        items.append(s.MakeBox((1000.0, -3.0, -2.0), (1001.5, -1.0, -0.5), red))                   # far from the origin
        items.append(s.MakeBox((0.0, -3.0, -2.0), (1.5, -3.0 + 1e-9, -0.5), green))                # paper-thin
        items.append(s.Translate(s.RotateY(s.MakeBox((0, 0, 0), (1.6, 2.8, 1.6), glass), 18.0), (2.5, -3.0, -3.0)))
        items.append(s.Translate(s.RotateY(s.MakeBox((0, 0, 0), (1.2, 1.2, 1.2), metal), -25.0), (-1.5, -3.0, -1.0)))
        items.append(s.Sphere((0.0, -1003.0, 0.0), 1000.0, grey))
        s.SetWorld(s.BvhNode(items) if world_kind == 0 else s.HittableList(items))
        s.Camera((0.5, 1.0, 9.0), (0.0, 0.0, -2.0), (0, 1, 0), 55.0, 96 / 64, 0.0, 10.0, 0.0, 1.0, (0.1, 0.1, 0.15))
        s.Commit()
    return build


@pytest.mark.parametrize("world_kind", [0, 1])
def test_quad_zoo_matches_the_oracle(world_kind):
    """Every (normal axis, u axis) pairing and winding of an axis-aligned quad, a slanted quad, boxes plain / far /
    paper-thin / instanced: the AAQuad and BoxRec shortcuts of flat_scene.h against R/Quad.h:52-99 evaluated in full
    by the oracle (not against the kernel's own general test)."""
    check(_quad_zoo(world_kind), w=96, h=64, spp=8)
