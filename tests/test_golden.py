"""Committed golden fixtures (tests/golden/oracle_golden.npz, made by tests/golden/make_golden.py).

They pin the ORACLE against regressions and give the GPU tests a checker that does not depend on the oracle
being rebuilt identically.  They are oracle outputs, not reference outputs (parity with the reference is
unpinned: it ships no vectors and cannot be built here)."""
import os

import numpy as np
import pytest

import raytracinginoneweekendincuda_amd as rt

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.npz"))
W, H, SPP, DEPTH, SEED = (int(x) for x in G["geometry"])


@pytest.mark.parametrize("scene", range(12))
def test_oracle_reproduces_golden_frames(oracle, scene):
    for world in (0, 1):
        fb, st = oracle.render(scene, world, W, H, SPP, depth=DEPTH, seed=SEED, earth=G["earth"], want_stats=True)
        want = G[f"frame_s{scene}_w{world}"]
        assert np.array_equal(fb.view(np.uint64), want.view(np.uint64))
        assert st["rays"] == int(G[f"rays_s{scene}_w{world}"][0])


def test_reference_invariant_bvh_equals_list_on_media_free_scenes():
    """Docs 2-3 BVH :733,:772 -- "MD5 identical with and without BVH".  Holds for every scene without a
    ConstantMedium (scenes 8 and 9 draw random numbers during traversal)."""
    for scene in (0, 1, 2, 3, 4, 5, 6, 7, 10, 11):
        a, b = G[f"frame_s{scene}_w0"], G[f"frame_s{scene}_w1"]
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), scene
    assert not np.array_equal(G["frame_s9_w0"], G["frame_s9_w1"])


def test_golden_rng_vectors(oracle):
    for k in range(5):
        seed, seq = (int(x) for x in G[f"rng_{k}_key"])
        raw, uni = oracle.rng_stream(seed, seq, 16)
        assert np.array_equal(raw, G[f"rng_{k}_raw"]) and np.array_equal(uni.view(np.uint32), G[f"rng_{k}_uniform"].view(np.uint32))
        r = rt.Rng(seed, seq)
        assert [r.next_u32() for _ in range(16)] == list(G[f"rng_{k}_raw"])


@pytest.mark.parametrize("scene", range(12))
def test_product_scene_tables_match_golden(scene):
    s = rt.builtin_scene(scene, 0, 1200, 800)
    kinds, boxes = s.dump_leaves()
    want_k = np.array([k if k <= 2 else 3 for k in G[f"leaves_s{scene}_kinds"]])
    assert np.array_equal(kinds, want_k)
    assert np.array_equal(boxes.view(np.uint64), G[f"leaves_s{scene}_boxes"].view(np.uint64))
    assert np.array_equal(s.dump_camera().view(np.uint64), G[f"camera_s{scene}"].view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("scene", range(12))
def test_gpu_strict_matches_golden_frames(scene):
    for world in (0, 1):
        s = rt.builtin_scene(scene, world, W, H, seed=SEED, earth=G["earth"])
        got, st = s.render(W, H, SPP, max_depth=DEPTH, seed=SEED, variant=0)
        want = G[f"frame_s{scene}_w{world}"]
        assert np.mean(np.all(np.abs(got - want) <= 1e-5, axis=-1)) >= 0.999
        assert st.rays == int(G[f"rays_s{scene}_w{world}"][0])
