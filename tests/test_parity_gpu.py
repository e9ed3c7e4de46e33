"""Parity of the HIP megakernel (through the C-ABI) against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): RNG bit-exact, colour within 1e-5.  The strict variant (no FMA
contraction) is additionally expected to be bit-identical to the oracle except where device libm
(pow/log/sin/acos/atan2) differs from glibc by an ulp; the bit-exact pixel fraction is asserted per scene.
"""
import numpy as np
import pytest

import raytracinginoneweekendincuda_amd as rt

pytestmark = pytest.mark.gpu

TOL = 1e-5  # absolute, on the gamma-corrected [0,1] colour the kernel stores (north_star: "fp32 colour within 1e-5")
W, H, SPP = 64, 32, 4

# (scene, world_kind, min fraction of bit-identical pixels for the strict variant)
CASES = [
    (10, 0, 0.99), (10, 1, 0.99),    # C1 three spheres
    (11, 1, 0.99),                   # C2 random spheres, list world
    (0, 0, 0.99), (0, 1, 0.99),      # C3 random spheres + moving spheres + checker
    # measured: scene 3 (marble: device sin) 0.960, scene 5 0.998, everything else 1.000 -- the floors sit a point or two below
    (1, 0, 0.99), (2, 0, 0.98), (3, 0, 0.94), (4, 0, 1.0), (5, 0, 0.98),
    (6, 0, 1.0), (7, 0, 1.0), (7, 1, 1.0),   # C4 Cornell + instances
    (8, 0, 0.98), (8, 1, 0.98),      # smoke
    (9, 0, 0.98), (9, 1, 0.98),      # C5 final scene
]


def compare(got, want):
    diff = np.abs(got - want)
    exact = np.mean(np.all(got.view(np.uint64) == want.view(np.uint64), axis=-1))
    within = np.mean(np.all(diff <= TOL, axis=-1))
    return exact, within, diff.max()


@pytest.mark.parametrize("scene_id,world_kind,min_exact", CASES)
def test_strict_matches_oracle(oracle, earth, scene_id, world_kind, min_exact):
    want = oracle.render(scene_id, world_kind, W, H, SPP, earth=earth)
    s = rt.builtin_scene(scene_id, world_kind, W, H, earth=earth)
    got, st = s.render(W, H, SPP, variant=0)
    exact, within, worst = compare(got, want)
    print(f"scene {scene_id} world {world_kind}: bit-exact {exact:.4f}, within {TOL:g}: {within:.4f}, max |d| {worst:.3g}")
    assert st.samples == W * H * SPP
    assert within >= 0.999, f"only {within:.4f} of pixels within {TOL}"
    assert exact >= min_exact


@pytest.mark.parametrize("scene_id,world_kind", [(10, 0), (11, 1), (0, 0), (7, 0), (8, 0), (9, 0)])
def test_fast_variant_within_tolerance(oracle, earth, scene_id, world_kind):
    want = oracle.render(scene_id, world_kind, W, H, SPP, earth=earth)
    got, _ = rt.builtin_scene(scene_id, world_kind, W, H, earth=earth).render(W, H, SPP, variant=1)
    exact, within, worst = compare(got, want)
    print(f"fast scene {scene_id}: bit-exact {exact:.4f}, within {within:.4f}, max |d| {worst:.3g}")
    assert within >= 0.995


def test_ray_counter_matches_oracle(oracle):
    want, stats = oracle.render(0, 0, W, H, SPP, want_stats=True)
    got, st = rt.builtin_scene(0, 0, W, H).render(W, H, SPP, variant=0)
    assert st.rays == stats["rays"]


@pytest.mark.parametrize("scene_id", [0, 3, 4, 7, 10])
def test_bvh_world_equals_list_world_bitwise(scene_id):
    """The reference's own invariant: BVH image == linear image, MD5-identical (Docs 2-3 :733,:772)."""
    a, _ = rt.builtin_scene(scene_id, 0, W, H).render(W, H, SPP, variant=0)
    b, _ = rt.builtin_scene(scene_id, 1, W, H).render(W, H, SPP, variant=0)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_stripe_partition_is_bit_identical_to_one_gpu(world):
    s = rt.builtin_scene(0, 0, 40, 50)
    full, _ = s.render(40, 50, 2, variant=0)
    parts = []
    rows_max = max(len(rt.stripe_rows(50, 8, r, world)) for r in range(world))
    for r in range(world):
        film = rt.Film(40, 50, stripe_rows=8, rank=r, world_size=world)
        film.render(s, 2, variant=0)
        mine = film.download()[rt.stripe_rows(50, 8, r, world)]
        buf = np.zeros(rows_max * 40 * 3)
        buf[: mine.size] = mine.ravel()
        parts.append(buf)
    out = rt.deinterleave(np.stack(parts), 40, 50, 8, world)
    assert np.array_equal(out.view(np.uint64), full.view(np.uint64))


@pytest.mark.parametrize("variant", [0, 1])
def test_instanced_list_kernel_on_five_waves_per_simd_gives_the_four_wave_frame(variant):
    """C4's world is scanned by one of two builds of the same kernel: four waves per SIMD, or five with the path state parked in
    LDS (render.hip Traits::PARK, list_instances_waves: whole generations of pixels on the resident lanes decide).  800 x 800
    pixels take the five-wave build, a quarter of the rows (stripes of four ranks) the four-wave one: same frame, same rays."""
    w = h = 800
    s = rt.builtin_scene(7, 0, w, h)
    film = rt.Film(w, h)
    st = film.render(s, 3, variant=variant)
    full = film.download().copy()
    assert st.kernel_vgprs <= 96, st.kernel_vgprs
    world, rays, parts = 4, 0, []
    rows_max = max(len(rt.stripe_rows(h, 8, r, world)) for r in range(world))
    for r in range(world):
        part = rt.Film(w, h, stripe_rows=8, rank=r, world_size=world)
        pst = part.render(s, 3, variant=variant)
        assert 96 < pst.kernel_vgprs <= 128, pst.kernel_vgprs
        rays += pst.rays
        mine = part.download()[rt.stripe_rows(h, 8, r, world)]
        buf = np.zeros(rows_max * w * 3)
        buf[: mine.size] = mine.ravel()
        parts.append(buf)
    out = rt.deinterleave(np.stack(parts), w, h, 8, world)
    assert np.array_equal(out.view(np.uint64), full.view(np.uint64))
    assert rays == st.rays


@pytest.mark.parametrize("scene_id,world_kind", [(11, 1), (0, 0)])
def test_a_ranks_stripes_of_the_benchmark_frame_equal_the_full_frames_rows(scene_id, world_kind):
    """One rank's share of an 8-way split of the 1200 x 800 frame holds fewer pixels than the GPU has lanes: its heavy pixels are
    served fewer to a wave than in the full frame (RenderArgs::adaptive_ppw, device_scene.cpp).  Same pixels as the full frame's."""
    w, h, spp, world, rank = 1200, 800, 64, 8, 3
    s = rt.builtin_scene(scene_id, world_kind, w, h)
    full = rt.Film(w, h)
    st_full = full.render(s, spp, variant=0)
    part = rt.Film(w, h, stripe_rows=8, rank=rank, world_size=world)
    st_part = part.render(s, spp, variant=0)
    rows = rt.stripe_rows(h, 8, rank, world)
    assert st_part.kernel_kind == st_full.kernel_kind
    assert np.array_equal(part.download()[rows].view(np.uint64), full.download()[rows].view(np.uint64))


def test_final_scene_rank_stripes_with_pixel_classes_equal_the_full_frames_rows(earth):
    """C5's kernel serves heavy and light pixels by wave where the frame is a few generations of pixels on the GPU's lanes -- one
    rank's stripes of an 8-way split -- and not for the whole frame (device_scene.cpp deep_roles).  Same pixels either way, and the
    media's random draws with them."""
    w, h, spp, world, rank = 1600, 1600, 64, 8, 4
    s = rt.builtin_scene(9, 0, w, h, earth=earth)
    full = rt.Film(w, h)
    st_full = full.render(s, spp, variant=0)
    part = rt.Film(w, h, stripe_rows=8, rank=rank, world_size=world)
    st_part = part.render(s, spp, variant=0)
    rows = rt.stripe_rows(h, 8, rank, world)
    assert st_part.kernel_kind == st_full.kernel_kind == 263
    assert np.array_equal(part.download()[rows].view(np.uint64), full.download()[rows].view(np.uint64))
    plain = rt.Film(w, h, stripe_rows=8, rank=rank, world_size=world)
    st_plain = plain.render(s, spp, variant=0, flags=64)   # RT_FLAG_NO_PIXEL_CLASSES
    assert st_plain.rays == st_part.rays
    assert np.array_equal(plain.download()[rows].view(np.uint64), part.download()[rows].view(np.uint64))


@pytest.mark.parametrize("scene_id,world_kind", [(11, 1), (0, 0), (9, 0)])
@pytest.mark.parametrize("w,h", [(1024, 512), (640, 416), (1992, 1000)])
def test_pixel_classes_on_other_frame_sizes(earth, scene_id, world_kind, w, h):
    """Heavy / light pixel classes (sphere lists in two tiers, the primitive BVH's one-to-a-wave list, the deep kernel's classes for
    frames of few generations, fewer pixels per serving wave where light pixels are few) at frame sizes that land in different
    bands of the launcher's rules: the frame, the ray count and the saved RNG streams equal those without classes."""
    s = rt.builtin_scene(scene_id, world_kind, w, h, earth=earth if scene_id == 9 else None)
    a, b = rt.Film(w, h), rt.Film(w, h)
    st_a = a.render(s, 64, variant=0)
    st_b = b.render(s, 64, variant=0, flags=64)   # RT_FLAG_NO_PIXEL_CLASSES
    assert st_a.rays == st_b.rays
    assert np.array_equal(a.download().view(np.uint64), b.download().view(np.uint64))
    st_a = a.render(s, 4, variant=0, flags=1)     # RT_FLAG_KEEP_RNG_STATE: four more samples from the saved streams
    st_b = b.render(s, 4, variant=0, flags=1 | 64)
    assert st_a.rays == st_b.rays
    assert np.array_equal(a.download().view(np.uint64), b.download().view(np.uint64))


def test_progressive_state_is_saved_and_resumed():
    """randState is written back (R/kernel.cu:146): 2 spp then 2 more spp continues the same streams."""
    s = rt.builtin_scene(10, 0, 32, 16)
    film = rt.Film(32, 16)
    film.render(s, 2, variant=0)
    a = film.download()
    film.launch(s, film.params(2, variant=0, flags=1))
    film.finish(s)
    b = film.download()
    four, _ = s.render(32, 16, 4, variant=0)
    # mean of the two halves (undo gamma) equals the 4-spp render up to the different summation order
    lin = (a ** 2 + b ** 2) / 2
    assert np.allclose(np.sqrt(lin), four, atol=1e-12)


def test_progressive_accumulation_equals_one_launch():
    """the reference saves randState but never reuses it.  Here 1 + 2 + 1 spp rendered progressively
    (saved streams + running sums) is bit-identical to one 4-spp launch: same draws, same summation order."""
    for scene_id, world in ((10, 0), (11, 1), (9, 0)):
        s = rt.builtin_scene(scene_id, world, W, H)
        film = rt.Film(W, H)
        film.render(s, 1, variant=0, flags=8)            # ACCUMULATE (first launch seeds)
        film.render(s, 2, variant=0, flags=8 | 1)        # ACCUMULATE | KEEP_RNG_STATE
        film.render(s, 1, variant=0, flags=8 | 1)
        one, _ = s.render(W, H, 4, variant=0)
        assert np.array_equal(film.download().view(np.uint64), one.view(np.uint64)), scene_id


def test_frames_in_flight_on_separate_films_are_independent():
    """Three films launched back to back on their own streams (frames overlap on the chip: the tail of one is filled by the
    next) give exactly the frames they give one at a time; each film's ray counter is its own."""
    jobs = [(10, 0, 3), (11, 1, 2), (7, 0, 2)]
    scenes = [rt.builtin_scene(sid, wk, W, H) for sid, wk, _ in jobs]
    films = [rt.Film(W, H) for _ in jobs]
    for s, f, (_, _, spp) in zip(scenes, films, jobs):
        f.launch(s, f.params(spp, variant=0))
    stats = [f.finish(s) for s, f in zip(scenes, films)]
    for s, f, st, (sid, _, spp) in zip(scenes, films, stats, jobs):
        alone, st_alone = s.render(W, H, spp, variant=0)
        assert np.array_equal(f.download().view(np.uint64), alone.view(np.uint64)), sid
        assert st.rays == st_alone.rays, sid


@pytest.mark.parametrize("scene_id", [0, 9])
def test_tile_ranking_does_not_change_the_image(scene_id, earth):
    """BVH worlds rehearse the first sample of every pixel, rank the 8x8 tiles by rays traced and start the heaviest
    first (the frame cannot end before its longest pixel).  The rehearsal writes nothing but tile costs -- the saved RNG
    streams are untouched -- so the frame equals the row-major one bit for bit; RT_FLAG_ROW_MAJOR_TILES (16) turns it off."""
    w = h = 256                                     # 1024 tiles: the smallest frame that is ranked
    s = rt.builtin_scene(scene_id, 0, w, h, earth=earth if scene_id == 9 else None)
    ranked, st_ranked = s.render(w, h, 32, variant=0)
    plain, st_plain = s.render(w, h, 32, variant=0, flags=16)
    assert np.array_equal(ranked.view(np.uint64), plain.view(np.uint64))
    assert st_ranked.rays == st_plain.rays          # the rehearsal's rays are not counted


def _quad_zoo(world_kind, plain):
    """Quads in every axis pairing and winding, a slanted quad, boxes plain / instanced / paper-thin / far from the origin."""
    if True:
        s = rt.Scene()
        s.set_options(rt.SCENE_PLAIN_QUADS if plain else 0)
        red, green, grey = s.Lambertian((0.65, 0.05, 0.05)), s.Lambertian((0.12, 0.45, 0.15)), s.Lambertian((0.73, 0.73, 0.73))
        metal, glass, light = s.Metal((0.8, 0.8, 0.9), 0.1), s.Dielectric(1.5), s.DiffuseLight((4.0, 4.0, 4.0))
        items = []
        e = [(1.5, 0, 0), (0, 1.5, 0), (0, 0, 1.5)]
        k = 0
        for a in range(3):                       # normal axis
            for p in range(3):                   # u axis, v on the remaining one: all six (a, p) pairings
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
        world = s.BvhNode(items) if world_kind == 0 else s.HittableList(items)
        s.SetWorld(world)
        s.Camera((0.5, 1.0, 9.0), (0.0, 0.0, -2.0), (0, 1, 0), 55.0, 96 / 64, 0.0, 10.0, 0.0, 1.0, (0.1, 0.1, 0.15))
        s.Commit()
        return s


@pytest.mark.parametrize("world_kind", [0, 1])
@pytest.mark.parametrize("variant", [0, 1])
def test_axis_aligned_quads_and_boxes_equal_the_general_quad_test(world_kind, variant):
    """flat_scene.h AAQuad / BoxRec: the one-multiply plane and interior tests, the six-planes-at-once box test and its
    inside / outside classification of the hit point give the same frame, bit for bit, as R/Quad.h:52-99 evaluated in
    full for every quad (RT_SCENE_PLAIN_QUADS at commit time).  Both builds: the shortcut drops exact zeros only."""
    fast = _quad_zoo(world_kind, plain=False)
    full = _quad_zoo(world_kind, plain=True)
    assert fast.info()["n_quads"] == full.info()["n_quads"]
    assert fast.info()["n_objects"] == 2 and full.info()["n_objects"] == 5   # plain boxes are leaves of their own (REF_BOX)
    a, sa = fast.render(96, 64, 8, variant=variant)
    b, sb = full.render(96, 64, 8, variant=variant)
    assert sa.rays == sb.rays
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    assert np.isfinite(a).all() and a.max() > 0.2


@pytest.mark.parametrize("scene_id", [1, 4, 6, 7, 10])
@pytest.mark.parametrize("variant", [0, 1])
def test_small_world_scan_equals_the_bvh_walk(scene_id, variant):
    """BVH worlds of up to 16 leaves without media are rendered by a scan of all leaves in the tree's leaf order (every
    lane on the same leaf, no node visits).  No leaf draws random numbers, so the closest hit is the one the walk finds:
    the frame equals the walked one (RT_FLAG_ALWAYS_WALK = 32) bit for bit, in both builds."""
    s = rt.builtin_scene(scene_id, 0, W, H)
    scan, st_scan = s.render(W, H, SPP, variant=variant)
    walk, st_walk = s.render(W, H, SPP, variant=variant, flags=32)
    assert st_scan.kernel_kind in (8, 10), "expected a list-scan instantiation"
    assert (st_walk.kernel_kind & 63) in (0, 2), "expected a BVH instantiation"
    assert st_scan.rays == st_walk.rays
    assert np.array_equal(scan.view(np.uint64), walk.view(np.uint64))


def test_full_size_rows_match_oracle(oracle):
    """Config C2 geometry (1200x800, list world): pixel RNG sequences depend on the full width, so check
    real rows of the full-size frame at low spp against the oracle."""
    Wf, Hf = 1200, 800
    s = rt.builtin_scene(11, 1, Wf, Hf)
    got, st = s.render(Wf, Hf, 1, variant=0)
    for row in (0, 399, 799):
        want = oracle.render(11, 1, Wf, Hf, 1, rows=(row, row + 1))
        exact, within, worst = compare(got[row:row + 1], want[row:row + 1])
        assert within >= 0.999 and exact >= 0.99, (row, exact, within, worst)


def test_ppm_bytes_match_oracle_writer(oracle, tmp_path):
    frame, _ = rt.builtin_scene(10, 0, 32, 16).render(32, 16, 2, variant=0)
    a, b = tmp_path / "a.ppm", tmp_path / "b.ppm"
    rt.write_ppm(a, frame)
    oracle.L.oracle_write_ppm(str(b).encode(), frame.ctypes.data, 32, 16)
    assert a.read_bytes() == b.read_bytes()
    assert a.read_bytes().startswith(b"P3\n32 16\n255\n")


# ---- scheduling variants must not change a single bit ----
def _render(scene_id, world, **kw):
    s = rt.builtin_scene(scene_id, world, W, H)
    return s.render(W, H, SPP, variant=0, **kw)


@pytest.mark.parametrize("scene_id,world_kind", [(11, 1), (10, 1)])
def test_cooperative_scan_equals_pixel_parallel_scan(scene_id, world_kind):
    """Sphere-list kernel: the ray-cooperative scan (64 lanes split one ray's spheres, DPP min over (t, k))
    returns the same hit as the per-lane sequential scan -- for every ray, not just in the frame tail."""
    ref, st0 = _render(scene_id, world_kind, coop_threshold=1, overdue=-1)       # never cooperative
    allc, st1 = _render(scene_id, world_kind, coop_threshold=65, overdue=-1)     # always cooperative
    over, st2 = _render(scene_id, world_kind, coop_threshold=1, overdue=1)       # pixels go cooperative after 1 ray/sample
    assert st0.kernel_kind == 16, "expected the sphere-list instantiation"
    assert np.array_equal(ref.view(np.uint64), allc.view(np.uint64))
    assert np.array_equal(ref.view(np.uint64), over.view(np.uint64))
    assert st0.rays == st1.rays == st2.rays
    for ppw in (1, 8, 24):   # a wave holds this many pixels and 64 / ppw lanes share each ray's scan
        few, st3 = _render(scene_id, world_kind, pixels_per_wave=ppw)
        assert np.array_equal(ref.view(np.uint64), few.view(np.uint64)), ppw
        assert st3.rays == st0.rays


def _many_spheres(n):
    s = rt.Scene()
    rng = np.random.default_rng(7)
    mats = [s.Lambertian((0.7, 0.3, 0.2)), s.Metal((0.8, 0.8, 0.9), 0.05), s.Dielectric(1.5), s.Lambertian((0.2, 0.5, 0.8))]
    items = [s.Sphere((0.0, -1000.0, 0.0), 1000.0, mats[0])]
    for k in range(n - 1):
        c = rng.uniform((-9, 0.15, -9), (9, 2.5, 9))
        items.append(s.Sphere(tuple(c), float(rng.uniform(0.1, 0.3)), mats[k % 4]))
    s.SetWorld(s.HittableList(items))
    s.Camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 30.0, W / H, 0.1, 10.0)
    s.Commit()
    return s


@pytest.mark.parametrize("n_spheres", [64, 1700])
def test_cooperative_scans_on_other_list_sizes(n_spheres):
    """64 spheres: one 64-row plane, every group width of the grouped scan sees a ragged tail of the table.
    1700 spheres: the sphere planes no longer fit the LDS budget, so thin waves fall back to the one-ray-at-a-time
    cooperative scan over the global table.  RT_FLAG_COOP_SINGLE forces that older scan on the LDS planes too."""
    s = _many_spheres(n_spheres)
    ref, st0 = s.render(W, H, SPP, variant=0, coop_threshold=1, overdue=-1)
    assert st0.kernel_kind == 16
    coop, st1 = s.render(W, H, SPP, variant=0, coop_threshold=65, overdue=-1)
    assert np.array_equal(ref.view(np.uint64), coop.view(np.uint64)) and st0.rays == st1.rays
    single, st2 = s.render(W, H, SPP, variant=0, coop_threshold=65, overdue=-1, flags=rt.FLAG_COOP_SINGLE)
    assert np.array_equal(ref.view(np.uint64), single.view(np.uint64)) and st0.rays == st2.rays


@pytest.mark.parametrize("scene_id,world_kind", [(11, 1), (0, 0), (4, 0), (7, 0), (8, 0)])
def test_specialised_kernels_equal_general_kernel(scene_id, world_kind):
    fast_path, st0 = _render(scene_id, world_kind)
    general, st1 = _render(scene_id, world_kind, flags=2)  # RT_FLAG_FORCE_GENERAL
    assert st0.kernel_kind != st1.kernel_kind
    assert np.array_equal(fast_path.view(np.uint64), general.view(np.uint64))


@pytest.mark.parametrize("batch", [1, 64])
def test_bvh_shading_batch_does_not_change_results(batch):
    a, _ = _render(0, 0)
    b, _ = _render(0, 0, shade_batch=batch)
    c, _ = _render(9, 0, shade_batch=batch)
    d, _ = _render(9, 0)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    assert np.array_equal(c.view(np.uint64), d.view(np.uint64))


def test_sub_bvh_group_equals_linear_group():
    """a large HittableList inside an instance gets a sub-BVH; 15 spheres stay a linear scan.
    Same spheres, two encodings -> same picture."""
    def build(n_extra):
        s = rt.Scene()
        white = s.Lambertian((0.73, 0.73, 0.73))
        rng = np.random.default_rng(5)
        pts = rng.uniform(0, 4, size=(15, 3))
        balls = [s.Sphere(tuple(p), 0.35, white) for p in pts]
        # far-away filler spheres (never hit) push the group over the sub-BVH threshold
        balls += [s.Sphere((1000.0 + 3 * k, 1000.0, 1000.0), 0.1, white) for k in range(n_extra)]
        grp = s.Translate(s.RotateY(s.HittableList(balls), 20.0), (-2.0, -2.0, -6.0))
        floor = s.Quad((-20, -3, -20), (40, 0, 0), (0, 0, 40), s.Lambertian((0.4, 0.6, 0.4)))
        s.SetWorld(s.HittableList([grp, floor]))
        s.Camera((0, 0, 3), (0, 0, -4), (0, 1, 0), 50, W / H, 0.0, 10.0)
        s.Commit()
        return s
    lin, few = build(0), build(40)
    assert lin.info()["n_nodes"] == 0 and few.info()["n_nodes"] > 0
    a, _ = lin.render(W, H, SPP, variant=0)
    b, _ = few.render(W, H, SPP, variant=0)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


# ---- edge cases ----
@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (9, 17), (65, 3)])
def test_ragged_frame_sizes(oracle, w, h):
    """Frames that are not multiples of the 8x8 tile (the reference's W/8+1 grid with a bounds check, Q21)."""
    for scene_id, world in ((10, 0), (11, 1), (7, 0)):
        want = oracle.render(scene_id, world, w, h, 2)
        got, st = rt.builtin_scene(scene_id, world, w, h).render(w, h, 2, variant=0)
        assert st.samples == w * h * 2
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (scene_id, w, h)


def test_depth_limits_and_zero_samples(oracle):
    s = rt.builtin_scene(10, 0, 24, 16)
    for depth in (1, 2, 3):
        want = oracle.render(10, 0, 24, 16, 3, depth=depth)
        got, _ = s.render(24, 16, 3, max_depth=depth, variant=0)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), depth
    want0 = oracle.render(10, 0, 24, 16, 3, depth=0)   # depth 0: RayColor returns black (accumulated), R/kernel.cu:71,97
    got0, st0 = s.render(24, 16, 3, max_depth=0, variant=0)
    assert np.array_equal(got0, want0) and st0.rays == 0
    film = rt.Film(24, 16)
    st = film.render(s, 0, variant=0)                    # 0 spp: nothing launched, nothing written
    assert st.rays == 0 and np.all(film.download() == 0)


def test_other_seeds_match_oracle(oracle):
    for seed in (1, 2**40 + 17):
        want = oracle.render(0, 0, 32, 16, 2, seed=seed)
        got, _ = rt.builtin_scene(0, 0, 32, 16, seed=seed).render(32, 16, 2, seed=seed, variant=0)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_single_leaf_bvh_with_medium_duplicated_leaf():
    """span-1 BvhNode: left == right == the same leaf (R/BvhNode.h:63-67).  A ConstantMedium leaf is therefore hit
    twice and draws twice; as a HittableList world it is hit once.  The two pictures must differ, and
    the BVH one must equal the general kernel's."""
    def build(world_kind):
        s = rt.Scene()
        ball = s.Sphere((0, 0, -3), 1.0, s.Dielectric(1.5))
        fog = s.ConstantMedium(ball, 0.8, (0.9, 0.2, 0.2))
        items = [fog]
        s.SetWorld(s.BvhNode(items) if world_kind == 0 else s.HittableList(items))
        s.Camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 60, W / H, 0.0, 10.0)
        s.Commit()
        return s
    bvh, _ = build(0).render(W, H, 8, variant=0)
    lst, _ = build(1).render(W, H, 8, variant=0)
    gen, _ = build(0).render(W, H, 8, variant=0, flags=2)
    assert np.array_equal(bvh.view(np.uint64), gen.view(np.uint64))
    assert not np.array_equal(bvh, lst)


def test_rtow_executable_and_rccl_gather_path(tmp_path):
    """The thin C++ host executable (the reference's main()): direct path, and the striped multi-GPU path with its
    RCCL gather (exercised here with a communicator of one), both byte-identical to the API's PPM."""
    import hashlib
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(rt.library_path()), "rtow")
    args = ["--scene", "10", "--width", "48", "--height", "32", "--spp", "3", "--variant", "strict"]
    a, b, c = tmp_path / "a.ppm", tmp_path / "b.ppm", tmp_path / "c.ppm"
    r = subprocess.run([exe, *args, "--output", str(a)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Rendering a 48x32 image with 3 samples per pixel in 8x8 blocks." in r.stderr and "Done. Saved to" in r.stderr
    r = subprocess.run([exe, *args, "--gpus", "1", "--output", str(b)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    frame, _ = rt.builtin_scene(10, 0, 48, 32).render(48, 32, 3, variant=0)
    rt.write_ppm(c, frame)
    digests = {hashlib.md5(p.read_bytes()).hexdigest() for p in (a, b, c)}
    assert len(digests) == 1
    bad = subprocess.run([exe, "--scene", "99"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 99  # the reference's checkCudaErrors exit code (R/kernel.cu:29-40)


# ---- the benchmark configurations at their real geometry ----
FULL = {"c2": (11, 1, 1200, 800), "c3": (0, 0, 1200, 800), "c4": (7, 0, 800, 800), "c5": (9, 0, 1600, 1600)}


def _stripe_of_full_frame(scene, w, h, stripe, spp, variant):
    """One 8-row stripe of the full frame through the stripe partition (rank = stripe index of as many ranks as stripes)."""
    film = rt.Film(w, h, stripe_rows=8, rank=stripe, world_size=(h + 7) // 8)
    st = film.render(scene, spp, variant=variant)
    return film.download()[stripe * 8:stripe * 8 + 8], st


@pytest.mark.parametrize("cfg", ["c3", "c4", "c5"])
def test_full_size_rows_match_oracle_bvh_configs(oracle, earth, cfg):
    """C3 (1200 wide), C4 (800x800: aspect 1.0, another view than the 2:1 test frames) and C5 (1600 wide, the stb-decoded
    earth texture): the pixel's RNG sequence is j*W+i, so real rows of the real frame, at 1 spp, strict build."""
    scene_id, world, w, h = FULL[cfg]
    e = earth if scene_id == 9 else None
    got, st = rt.builtin_scene(scene_id, world, w, h, earth=e).render(w, h, 1, variant=0)
    for row in (0, h // 2 - 1, h - 1):
        want = oracle.render(scene_id, world, w, h, 1, earth=e, rows=(row, row + 1))
        exact, within, worst = compare(got[row:row + 1], want[row:row + 1])
        print(f"{cfg} row {row}: bit-exact {exact:.4f}, within {within:.4f}, max |d| {worst:.3g}")
        assert within >= 0.999 and exact >= (0.95 if cfg == "c5" else 0.99), (cfg, row, exact, within, worst)


# (config, spp, min within-1e-5 for the strict build, for the fast build, min bit-exact for the strict build)
BANDS = [("c2", 500, 0.999, 0.99, 0.99), ("c3", 500, 0.999, 0.99, 0.99), ("c4", 1000, 0.999, 0.99, 0.99),
         ("c5", 64, 0.999, 0.85, 0.90)]   # fast build: one contracted comparison that falls the other way re-draws the rest of the pixel's
                                          # stream; 0.91 of the pixels stay within tolerance at 64 spp (0.75 at 5000: bench.py times strict)


@pytest.mark.parametrize("cfg,spp,min_strict,min_fast,min_exact", BANDS)
def test_band_of_the_benchmark_frame_at_benchmark_spp(oracle, earth, cfg, spp, min_strict, min_fast, min_exact):
    """Eight full-width rows through the middle of the benchmark frame at the benchmark's own spp (C5: 64 of its 5000),
    both builds, against the oracle.  This is where a contracted discriminant flipping a grazing hit would show: a flip
    moves a pixel by ~1/spp of a colour.  (bench.py reports the same comparison for the frame it times: `parity`.)"""
    scene_id, world, w, h = FULL[cfg]
    e = earth if scene_id == 9 else None
    stripe = (h // 2) // 8
    want = oracle.render(scene_id, world, w, h, spp, earth=e, rows=(stripe * 8, stripe * 8 + 8))[stripe * 8:stripe * 8 + 8]
    scene = rt.builtin_scene(scene_id, world, w, h, earth=e)
    for variant, floor in ((0, min_strict), (1, min_fast)):
        got, _ = _stripe_of_full_frame(scene, w, h, stripe, spp, variant)
        exact, within, worst = compare(got, want)
        q = lambda f: (256.0 * np.clip(f, 0.0, 0.999)).astype(np.int32)
        ppm = np.mean(np.all(q(got) == q(want), axis=-1))
        print(f"{cfg} x{spp}spp {'strict' if variant == 0 else 'fast'}: bit-exact {exact:.4f}, within {TOL:g}: {within:.4f}, "
              f"8-bit PPM equal {ppm:.4f}, max |d| {worst:.3g}")
        assert within >= floor, (cfg, variant, within)
        if variant == 0:
            assert exact >= min_exact, (cfg, exact)


def test_scene_destroyed_while_its_launch_is_in_flight():
    """rt_scene_destroy with a launch in flight waits for the kernel (it reads the scene's tables) and detaches the film:
    the film's finish / download still deliver the frame, and a film destroyed unfinished gives the scene its count back."""
    import ctypes as C
    L = rt.lib()
    w, h, spp = 256, 128, 16
    want, _ = rt.builtin_scene(0, 0, w, h).render(w, h, spp, variant=0)
    scene = rt.builtin_scene(0, 0, w, h)
    film = rt.Film(w, h)
    film.launch(scene, film.params(spp, variant=0))
    film._scene = None                # what Film.launch keeps alive on purpose: drop it, then the scene itself
    L.rt_scene_destroy(scene._p)
    scene._p = None
    st = film.finish()
    assert st.samples == w * h * spp
    assert np.array_equal(film.download().view(np.uint64), want.view(np.uint64))
    # a film that goes away with its launch unfinished: the scene can be changed again afterwards
    scene2 = rt.builtin_scene(0, 0, w, h)
    film2 = rt.Film(w, h)
    film2.launch(scene2, film2.params(spp, variant=0))
    with pytest.raises(rt.RtowError):
        scene2.Camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)   # in flight: refused
    L.rt_film_destroy(film2._p)
    film2._p = None
    scene2.Camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, w / h, 0.1, 10.0)


def test_c5_rows_at_its_real_5000_spp(oracle, earth):
    """Config C5 at the spp BASELINE.json quotes it on: two full-width rows through the middle of the 1600x1600 frame, 5000
    samples per pixel (8 M samples, ~22 M rays through the mist, the glass and the marble), strict build -- the build bench.py
    times for C5 -- against the oracle.  Every pixel's 5000 samples are one RNG stream: a single decision that fell the other
    way anywhere in it would move the pixel by more than the tolerance, so `within` is the test; bit-exactness is lost only
    to device sin / log / acos differing from glibc's by an ulp in a value that is then averaged (measured 0.995).
    (The fast build is NOT expected to pass this: see rt_render_params.variant in include/rtow.h.)"""
    scene_id, world, w, h = FULL["c5"]
    stripe, n_rows, spp = (h // 2) // 8, 2, 5000
    want = oracle.render(scene_id, world, w, h, spp, earth=earth, rows=(stripe * 8, stripe * 8 + n_rows))[stripe * 8:stripe * 8 + n_rows]
    scene = rt.builtin_scene(scene_id, world, w, h, earth=earth)
    got, st = _stripe_of_full_frame(scene, w, h, stripe, spp, 0)
    exact, within, worst = compare(got[:n_rows], want)
    print(f"c5 x{spp}spp strict, rows {stripe * 8}..{stripe * 8 + n_rows - 1}: bit-exact {exact:.4f}, within {TOL:g}: {within:.4f}, "
          f"max |d| {worst:.3g}, {st.rays / st.samples:.2f} rays/sample")
    assert within >= 0.9995 and exact >= 0.98, (exact, within, worst)
    # the same two rows as rank 4 of an 8-way split renders them -- 200 rows, 1.6 generations of pixels on the GPU's lanes, so
    # with heavy and light pixel classes (device_scene.cpp deep_roles): the stripe render's pixels bit for bit
    part = rt.Film(w, h, stripe_rows=8, rank=stripe % 8, world_size=8)
    part.render(scene, spp, variant=0)
    rows = part.download()[stripe * 8:stripe * 8 + n_rows]
    assert np.array_equal(rows.view(np.uint64), got[:n_rows].view(np.uint64))


# ---- a scene that changes between renders (frame sequences) ----
def test_scene_changed_after_a_render_is_uploaded_again():
    """Render, move the camera, add a sphere, commit, render again: the second frame must be the new scene's, bit for
    bit the frame of a scene built that way from scratch (the device tables are versioned, see SceneImpl::generation)."""
    def build(s, extra, cam_x):
        items = [s.Sphere((0.0, -100.5, -1.0), 100.0, s.Lambertian((0.8, 0.8, 0.0))),
                 s.Sphere((0.0, 0.0, -1.2), 0.5, s.Lambertian((0.1, 0.2, 0.5)))]
        if extra:
            items.append(s.Sphere((1.0, 0.0, -1.0), 0.5, s.Metal((0.8, 0.6, 0.2), 0.3)))
        s.SetWorld(s.BvhNode(items))
        s.Camera((cam_x, 0, 0.5), (0, 0, -1), (0, 1, 0), 70.0, W / H, 0.0, 10.0)
        s.Commit()

    s = rt.Scene()
    build(s, False, 0.0)
    first, _ = s.render(W, H, SPP, variant=0)
    # only the camera moves: no commit needed, the tables are unchanged
    s.Camera((0.7, 0, 0.5), (0, 0, -1), (0, 1, 0), 70.0, W / H, 0.0, 10.0)
    moved, _ = s.render(W, H, SPP, variant=0)
    fresh = rt.Scene()
    build(fresh, False, 0.7)
    want_moved, _ = fresh.render(W, H, SPP, variant=0)
    assert not np.array_equal(first, moved)
    assert np.array_equal(moved.view(np.uint64), want_moved.view(np.uint64))
    # new world + commit
    build(s, True, 0.7)
    third, _ = s.render(W, H, SPP, variant=0)
    fresh2 = rt.Scene()
    build(fresh2, True, 0.7)
    want_third, _ = fresh2.render(W, H, SPP, variant=0)
    assert np.array_equal(third.view(np.uint64), want_third.view(np.uint64))
    assert not np.array_equal(third, moved)


def test_state_errors_while_a_render_is_in_flight():
    s = rt.builtin_scene(11, 1, 256, 128)
    film = rt.Film(256, 128)
    film.launch(s, film.params(64, variant=0))
    with pytest.raises(rt.RtowError):     # a film holds one frame in flight
        film.launch(s, film.params(1, variant=0))
    with pytest.raises(rt.RtowError):     # the kernel may still be reading the tables
        s.Camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 30.0, 2.0, 0.1, 10.0)
    with pytest.raises(rt.RtowError):
        s.Commit()
    st = film.finish(s)
    assert st.samples == 256 * 128 * 64
    s.Camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 30.0, 2.0, 0.1, 10.0)   # fine again
    film.render(s, 1, variant=0)


def test_rtow_executable_takes_the_earth_texture(earth, tmp_path):
    """Scene 9 is the reference's default and needs earthmap.jpg (R/kernel.cu:656-665).  The executable takes the decoded
    image as a P6 PPM; with the stb-decoded fixture its output.ppm equals the API's render with the same bytes, and
    without one it says so and renders the reference's own missing-file fallback (cyan)."""
    import hashlib
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(rt.library_path()), "rtow")
    ppm = tmp_path / "earth_bytes.ppm"
    with open(ppm, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (earth.shape[1], earth.shape[0]))
        f.write(earth.tobytes())
    args = ["--scene", "9", "--width", "64", "--height", "48", "--spp", "2", "--variant", "strict"]
    a, b, c = tmp_path / "a.ppm", tmp_path / "b.ppm", tmp_path / "c.ppm"
    r = subprocess.run([exe, *args, "--earth-bytes", str(ppm), "--output", str(a)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Loaded image" in r.stderr and "(1024x512)" in r.stderr
    frame, _ = rt.builtin_scene(9, 0, 64, 48, earth=earth).render(64, 48, 2, variant=0)
    rt.write_ppm(b, frame)
    assert hashlib.md5(a.read_bytes()).hexdigest() == hashlib.md5(b.read_bytes()).hexdigest()
    r = subprocess.run([exe, *args, "--output", str(c)], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 0 and "Could not load image file" in r.stderr
    cyan, _ = rt.builtin_scene(9, 0, 64, 48).render(64, 48, 2, variant=0)
    rt.write_ppm(b, cyan)
    assert hashlib.md5(c.read_bytes()).hexdigest() == hashlib.md5(b.read_bytes()).hexdigest()


def test_rtow_executable_list_world_flags(tmp_path):
    """`rtow --world list` scans the list as the reference does; `--accelerate-lists` (RT_FLAG_ACCELERATE_LISTS) and `--flags`
    (here RT_FLAG_EXACT_SCAN = 256) change how it is searched, never the picture: three identical output files."""
    import hashlib
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(rt.library_path()), "rtow")
    args = ["--scene", "11", "--world", "list", "--width", "160", "--height", "96", "--spp", "8", "--variant", "strict"]
    digests = []
    for k, extra in enumerate(([], ["--accelerate-lists"], ["--flags", "256"])):
        out = tmp_path / f"o{k}.ppm"
        r = subprocess.run([exe, *args, *extra, "--output", str(out)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        digests.append(hashlib.md5(out.read_bytes()).hexdigest())
    assert digests[0] == digests[1] == digests[2]
    frame, _ = rt.builtin_scene(11, 1, 160, 96).render(160, 96, 8, variant=0)
    ref = tmp_path / "api.ppm"
    rt.write_ppm(ref, frame)
    assert hashlib.md5(ref.read_bytes()).hexdigest() == digests[0]


@pytest.mark.parametrize("scene_id", [0, 11])
@pytest.mark.parametrize("variant", [0])
def test_thin_wave_scan_of_a_sphere_bvh_world_equals_the_walk(oracle, scene_id, variant):
    """BVH worlds of spheres / moving spheres (config C3): once the pixel queue is dry, a wave with few live lanes stops
    walking and scans every leaf cooperatively (scan_grouped_ms).  No leaf draws random numbers, so the closest hit is the
    walk's: same frame bit for bit, same ray count -- with the switch forced on as early as possible (threshold 65), with
    the default, and never (threshold 1); and the scanned frame equals the oracle's."""
    w, h, spp = 96, 64, 6
    s = rt.builtin_scene(scene_id, 0, w, h)
    walk, st0 = s.render(w, h, spp, variant=variant, coop_threshold=1)
    scan, st1 = s.render(w, h, spp, variant=variant, coop_threshold=65)
    dflt, st2 = s.render(w, h, spp, variant=variant)
    assert (st0.kernel_kind & 63) == 0, "expected the primitive BVH instantiation"
    assert st0.rays == st1.rays == st2.rays
    assert np.array_equal(walk.view(np.uint64), scan.view(np.uint64))
    assert np.array_equal(walk.view(np.uint64), dflt.view(np.uint64))
    want = oracle.render(scene_id, 0, w, h, spp)
    assert np.array_equal(scan.view(np.uint64), want.view(np.uint64))


def test_heavy_and_light_pixels_in_two_launches_give_the_same_frame(oracle):
    """Sphere-list worlds (config C2): a rehearsal of the first samples finds the pixels with long ray chains (glass);
    they are rendered by a launch of their own, a few pixels per wave with the lanes sharing each ray's scan, beside the
    launch of all the others.  Every pixel is rendered exactly once from its own RNG stream, so the frame, the ray count
    and the saved RNG state are those of the single launch (RT_FLAG_NO_PIXEL_CLASSES = 64); rows against the oracle too."""
    w, h, spp = 512, 256, 64          # 131072 pixels: the smallest frame that is split
    s = rt.builtin_scene(11, 1, w, h)
    film_a, film_b = rt.Film(w, h), rt.Film(w, h)
    st_a = film_a.render(s, spp, variant=0)
    st_b = film_b.render(s, spp, variant=0, flags=64)
    a, b = film_a.download(), film_b.download()
    assert st_a.kernel_kind == 16
    assert st_a.rays == st_b.rays and st_a.samples == st_b.samples == w * h * spp
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    want = oracle.render(11, 1, w, h, spp, rows=(96, 100))
    assert np.array_equal(a[96:100].view(np.uint64), want[96:100].view(np.uint64))
    # progressive continuation from the saved streams: both films carry on identically
    film_a.render(s, 8, variant=0, flags=1)
    film_b.render(s, 8, variant=0, flags=1 | 64)
    assert np.array_equal(film_a.download().view(np.uint64), film_b.download().view(np.uint64))
    # the benchmark's frame size: its rehearsal finds some fifty pixels of 30 rays per sample and more, which the serving waves
    # take one to a wave before anything else (RenderArgs::super_list)
    w, h, spp = 1200, 800, 64
    s = rt.builtin_scene(0, 0, w, h)
    film_a, film_b = rt.Film(w, h), rt.Film(w, h)
    st_a = film_a.render(s, spp, variant=0)
    st_b = film_b.render(s, spp, variant=0, flags=64)
    assert st_a.kernel_kind == 64 and st_a.rays == st_b.rays
    assert np.array_equal(film_a.download().view(np.uint64), film_b.download().view(np.uint64))


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("scene_id,shape", [(11, (512, 256, 32)), (10, (160, 96, 24)), (1, (96, 64, 16)), (11, (64, 40, 16))])
def test_filtered_list_scan_gives_the_exact_scans_frame(oracle, scene_id, shape, variant):
    """Sphere-list worlds: every ray examines every sphere, but through an 8-instruction conservative filter
    (render.hip filter_four) instead of the reference's 13-instruction discriminant; only the spheres the filter cannot
    rule out go through the reference's arithmetic.  A filter that never rejects what the reference accepts changes
    nothing: the frame, the ray count and the RNG streams are those of the exact scan (RT_FLAG_EXACT_SCAN = 256) bit for
    bit, in both builds, for the pixel-parallel scan, the heavy-pixel groups and the frame-tail cooperative scan; the
    strict build's rows equal the oracle's."""
    w, h, spp = shape
    s = rt.builtin_scene(scene_id, 1, w, h)
    film_a, film_b = rt.Film(w, h), rt.Film(w, h)
    st_a = film_a.render(s, spp, variant=variant)
    st_b = film_b.render(s, spp, variant=variant, flags=rt.FLAG_EXACT_SCAN)
    assert st_a.kernel_kind == 16
    assert st_a.rays == st_b.rays
    a, b = film_a.download(), film_b.download()
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    # the default filter is the packed fp32 form (two spheres per instruction); the fp64 form is the same frame again
    film_c = rt.Film(w, h)
    st_c = film_c.render(s, spp, variant=variant, flags=rt.FLAG_FILTER_FP64)
    assert st_c.rays == st_a.rays and np.array_equal(film_c.download().view(np.uint64), a.view(np.uint64))
    if variant == 0:
        rows = (h // 2, h // 2 + 2)
        want = oracle.render(scene_id, 1, w, h, spp, rows=rows)
        assert np.array_equal(a[rows[0]:rows[1]].view(np.uint64), want[rows[0]:rows[1]].view(np.uint64))
    film_a.render(s, 4, variant=variant, flags=1)
    film_b.render(s, 4, variant=variant, flags=1 | rt.FLAG_EXACT_SCAN)
    assert np.array_equal(film_a.download().view(np.uint64), film_b.download().view(np.uint64))
    # a few pixels per wave: every scan of the frame is a grouped scan
    g1, st1 = s.render(w, h, 4, variant=variant, pixels_per_wave=8)
    g2, st2 = s.render(w, h, 4, variant=variant, pixels_per_wave=8, flags=rt.FLAG_EXACT_SCAN)
    assert st1.rays == st2.rays and np.array_equal(g1.view(np.uint64), g2.view(np.uint64))


@pytest.mark.parametrize("scene_id,shape", [(11, (512, 256, 16)), (0, (160, 96, 8)), (10, (160, 96, 8)), (4, (96, 64, 8))])
def test_list_world_through_the_library_tree(oracle, scene_id, shape):
    """RT_FLAG_ACCELERATE_LISTS: a HittableList world of primitives rendered through the library's tree, as a BvhNode world of
    the same objects would be -- the reference's "BVH image == list image" invariant (Docs 2-3 BVH :733,:772) used the other
    way round.  Same frame, ray count and continued RNG streams as the list scan, hence as the oracle's list world."""
    w, h, spp = shape
    s = rt.builtin_scene(scene_id, 1, w, h)
    film_a, film_b = rt.Film(w, h), rt.Film(w, h)
    st_a = film_a.render(s, spp, variant=0)
    st_b = film_b.render(s, spp, variant=0, flags=rt.FLAG_ACCELERATE_LISTS)
    a, b = film_a.download(), film_b.download()
    print(f"scene {scene_id}: list kernel kind {st_a.kernel_kind}, accelerated kind {st_b.kernel_kind}")
    if scene_id != 4:   # five quads: fewer than the three leaves... the tree is built from three leaves up; small worlds may stay lists
        assert st_b.kernel_kind & 64, "the flag did not select the library-tree kernel"
    assert st_a.rays == st_b.rays
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    rows = (h // 2, h // 2 + 2)
    want = oracle.render(scene_id, 1, w, h, spp, rows=rows)
    assert np.array_equal(b[rows[0]:rows[1]].view(np.uint64), want[rows[0]:rows[1]].view(np.uint64))
    film_a.render(s, 4, variant=0, flags=1)
    film_b.render(s, 4, variant=0, flags=1 | rt.FLAG_ACCELERATE_LISTS)
    assert np.array_equal(film_a.download().view(np.uint64), film_b.download().view(np.uint64))


@pytest.mark.parametrize("scene_id", [0, 11, 3])
@pytest.mark.parametrize("variant", [0])
def test_library_tree_gives_the_reference_trees_frame(oracle, scene_id, variant):
    """BVH worlds of primitives only are walked through the library's own tree (surface-area heuristic, near child first by
    the ray's direction octant; flat_scene.h FastNodeRec).  No leaf draws random numbers, so the closest hit -- hence the
    frame, the ray count and the RNG streams -- are those of the reference's median-split tree walked in the reference's
    order (RT_FLAG_REFERENCE_TREE = 128), which in turn equal the oracle's."""
    w, h, spp = 96, 64, 6
    s = rt.builtin_scene(scene_id, 0, w, h)
    own, st_own = s.render(w, h, spp, variant=variant)
    ref, st_ref = s.render(w, h, spp, variant=variant, flags=128)
    if scene_id != 3:   # scene 3 (two marble spheres) has table textures: the general kernel walks the reference's tree
        assert st_own.kernel_kind == 64 and st_ref.kernel_kind == 0
    assert st_own.rays == st_ref.rays
    assert np.array_equal(own.view(np.uint64), ref.view(np.uint64))
    if scene_id != 3:
        want = oracle.render(scene_id, 0, w, h, spp)
        assert np.array_equal(own.view(np.uint64), want.view(np.uint64))


def test_library_tree_on_a_mixed_primitive_world():
    """Spheres, moving spheres and quads of very different sizes in one BvhNode world (a huge ground sphere, a quad wall,
    thin and overlapping leaves): the library's tree against the reference's, both builds."""
    def build(s):
        rng = np.random.default_rng(11)
        mats = [s.Lambertian((0.7, 0.3, 0.2)), s.Metal((0.8, 0.8, 0.9), 0.05), s.Dielectric(1.5), s.Lambertian((0.2, 0.5, 0.8))]
        items = [s.Sphere((0.0, -1000.0, 0.0), 1000.0, mats[0]),
                 s.Quad((-6, 0, -7), (12, 0, 0), (0, 5, 0), mats[1]),
                 s.Quad((-6, 0, -7), (0, 0, 9), (0, 4, 0), mats[3])]
        for k in range(150):
            c = rng.uniform((-6, 0.15, -6), (6, 2.5, 4))
            r = float(rng.uniform(0.1, 0.5))
            if k % 3 == 0:
                items.append(s.MovingSphere(tuple(c), tuple(c + (0, 0.4, 0)), 0.0, 1.0, r, mats[k % 4]))
            else:
                items.append(s.Sphere(tuple(c), r, mats[k % 4]))
        s.SetWorld(s.BvhNode(items))
        s.Camera((10, 3, 8), (0, 1, -1), (0, 1, 0), 35.0, W / H, 0.05, 10.0, 0.0, 1.0)
        s.Commit()
        return s
    s = build(rt.Scene())
    for variant in (0, 1):
        own, st_own = s.render(W, H, SPP, variant=variant)
        ref, st_ref = s.render(W, H, SPP, variant=variant, flags=128)
        assert st_own.kernel_kind == 64 and st_ref.kernel_kind == 0
        assert st_own.rays == st_ref.rays
        assert np.array_equal(own.view(np.uint64), ref.view(np.uint64))


def test_heavy_pixel_server_waves_give_the_same_frame(oracle):
    """Primitive BVH worlds on the library's tree (config C3): the kernel's 768-thread workgroup fills a CU, so the pixels
    with long ray chains are not given a launch of their own but two waves of every workgroup, which take them off the
    list a few at a time and join the tile queue afterwards; the tile queue skips them.  Same frame, rays and streams as
    without classes (RT_FLAG_NO_PIXEL_CLASSES), and as the oracle."""
    w, h, spp = 512, 256, 64
    s = rt.builtin_scene(0, 0, w, h)
    film_a, film_b = rt.Film(w, h), rt.Film(w, h)
    st_a = film_a.render(s, spp, variant=0)
    st_b = film_b.render(s, spp, variant=0, flags=64)
    a, b = film_a.download(), film_b.download()
    assert st_a.kernel_kind == 64
    assert st_a.rays == st_b.rays
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    want = oracle.render(0, 0, w, h, spp, rows=(120, 124))
    assert np.array_equal(a[120:124].view(np.uint64), want[120:124].view(np.uint64))
    film_a.render(s, 8, variant=0, flags=1)
    film_b.render(s, 8, variant=0, flags=1 | 64)
    assert np.array_equal(film_a.download().view(np.uint64), film_b.download().view(np.uint64))
