"""The C-ABI library loads and exports exactly what include/rtow.h declares (no compute calls)."""
import os
import re

import raytracinginoneweekendincuda_amd as rt
from raytracinginoneweekendincuda_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtow.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"RTOW_API[^;(]*?\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 50
    L = rt.lib()
    for s in syms:
        assert hasattr(L, s), f"{s} declared in rtow.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "python binding table and rtow.h disagree"


def test_version_and_error_strings():
    assert b"gfx950" in rt.lib().rt_version()
    s = rt.Scene()
    try:
        s.Sphere((0, 0, 0), 1.0, 12345)  # invalid material handle
    except rt.RtowError as e:
        assert "material" in str(e)
    else:
        raise AssertionError("invalid handle accepted")


def test_commit_requires_world_and_camera():
    s = rt.Scene()
    m = s.Lambertian((0.5, 0.5, 0.5))
    sp = s.Sphere((0, 0, -1), 0.5, m)
    try:
        s.Commit()
    except rt.RtowError as e:
        assert "world" in str(e)
    else:
        raise AssertionError
    s.SetWorld(s.HittableList([sp]))
    try:
        s.Commit()
    except rt.RtowError as e:
        assert "camera" in str(e)
    else:
        raise AssertionError


def test_general_nesting_commits_and_only_absurd_depth_is_refused():
    """The reference's wrappers take any Hittable*: a medium inside a medium, a medium under a transform, a BvhNode inside
    a list all commit (kept as an object tree for the nested kernels).  What is refused is a nesting deeper than the
    interpreter's stack -- the reference's own recursion is bounded too (32 KiB of stack per thread, R/kernel.cu:599)."""
    s = rt.Scene()
    m = s.Lambertian((0.5, 0.5, 0.5))
    inner = s.ConstantMedium(s.Sphere((0, 0, 0), 1.0, m), 0.1, (1, 1, 1))
    outer = s.ConstantMedium(inner, 0.1, (1, 1, 1))
    moved = s.Translate(s.ConstantMedium(s.Sphere((3, 0, 0), 1.0, m), 0.2, (1, 1, 1)), (0, 1, 0))
    tree = s.BvhNode([s.MakeBox((5, 0, 0), (6, 1, 1), m), s.ConstantMedium(s.Sphere((8, 0, 0), 1.0, m), 0.3, (1, 1, 1))])
    s.SetWorld(s.HittableList([outer, moved, tree]))
    s.Camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 1.0, 0.0, 10.0)
    s.Commit()
    assert s.info()["n_media"] == 4 and s.info()["n_leaves"] == 3

    deep = rt.Scene()
    obj = deep.ConstantMedium(deep.Sphere((0, 0, 0), 1.0, deep.Lambertian((0.5, 0.5, 0.5))), 0.1, (1, 1, 1))
    for k in range(20):   # lists of one are frames of their own
        obj = deep.HittableList([deep.Translate(obj, (0.1, 0, 0))])
    deep.SetWorld(deep.HittableList([obj]))
    deep.Camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 1.0, 0.0, 10.0)
    try:
        deep.Commit()
    except rt.RtowError as e:
        assert "nesting deeper" in str(e)
    else:
        raise AssertionError


def test_render_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    s = rt.builtin_scene(10, 0, 16, 8)
    try:
        s.render(16, 8, 1)
    except rt.RtowError as e:
        assert "no HIP device" in str(e) or "HIP error" in str(e)
    else:
        raise AssertionError("render succeeded without a GPU: a CPU fallback must not exist")


def test_rtwimage_byte_conversion_matches_reference_formula():
    """FloatToByte(pow(b/255, 2.2)) with the float/double mix of stb + RtwImage (R/RtwImage.h:100-105)."""
    import numpy as np
    src = np.arange(256, dtype=np.uint8).reshape(1, 256, 1).repeat(3, axis=2)
    got = rt.rtwimage_bytes(src)[0, :, 0]
    v = (np.float32(np.arange(256)) / np.float32(255.0)).astype(np.float64) ** np.float64(np.float32(2.2))
    v = v.astype(np.float32)
    want = np.where(v <= 0, 0, np.where(v >= 1, 255, (np.float32(256.0) * v).astype(np.uint8)))
    assert np.array_equal(got, want.astype(np.uint8))
    assert got[0] == 0 and got[255] == 255 and got[128] == 56 and np.all(np.diff(got.astype(int)) >= 0)


def test_binary_ppm_and_pfm_writers(tmp_path):
    import numpy as np
    rng = np.random.default_rng(3)
    frame = rng.uniform(-0.1, 1.2, size=(5, 7, 3))
    p3, p6, pf = tmp_path / "a.ppm", tmp_path / "b.ppm", tmp_path / "c.pfm"
    rt.write_ppm(p3, frame); rt.write_ppm_binary(p6, frame); rt.write_pfm(pf, frame)
    text = p3.read_text().split()
    assert text[:4] == ["P3", "7", "5", "255"]
    vals = np.array(text[4:], dtype=np.int64)
    raw = p6.read_bytes()
    assert raw.startswith(b"P6\n7 5\n255\n") and np.array_equal(np.frombuffer(raw[len(b"P6\n7 5\n255\n"):], dtype=np.uint8), vals)
    want = (256.0 * np.clip(frame[::-1], 0.0, 0.999)).astype(np.int64).ravel()   # top row first, clamp, truncate
    assert np.array_equal(vals, want)
    body = pf.read_bytes()
    assert body.startswith(b"PF\n7 5\n-1.0\n")
    assert np.array_equal(np.frombuffer(body[len(b"PF\n7 5\n-1.0\n"):], dtype="<f4"), frame.astype(np.float32).ravel())


def test_python_flag_constants_match_the_header():
    import re
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "rtow.h")).read()
    flags = re.findall(r"#define RT_FLAG_(\w+) (\d+)u", header)
    assert len(flags) >= 5
    for name, value in flags:
        assert getattr(rt, "FLAG_" + name) == int(value), name
