import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


N_STATS = 13
STAT_NAMES = ["rays", "box_tests", "sphere_tests", "msphere_tests", "quad_tests", "xform_entries", "medium_calls",
              "medium_draws", "list_entries", "scatters", "noise_calls", "image_lookups", "rng_draws"]


class Oracle:
    """ctypes view of oracle/liboracle.so -- the CPU restatement used as the checker (tests only)."""

    def __init__(self):
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(path)
        L.oracle_render.argtypes = [C.c_int] * 6 + [C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                     C.c_void_p, C.c_void_p]
        L.oracle_render.restype = C.c_int
        L.oracle_scene_dump.argtypes = [C.c_int] * 4 + [C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_scene_dump.restype = C.c_int
        L.oracle_rng_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_rng_state.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
        L.oracle_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.oracle_write_ppm.restype = C.c_long
        assert L.oracle_stats_fields() == N_STATS
        self.L = L

    def render(self, scene_id, world_kind, W, H, spp, depth=50, seed=1984, earth=None, rows=None, threads=0,
               want_stats=False):
        fb = np.zeros((H, W, 3), dtype=np.float64)
        st = (C.c_uint64 * N_STATS)()
        r0, r1 = rows if rows else (0, H)
        ep, ew, eh = (None, 0, 0)
        if earth is not None:
            earth = np.ascontiguousarray(earth, dtype=np.uint8)
            ep, ew, eh = earth.ctypes.data, earth.shape[1], earth.shape[0]
        rc = self.L.oracle_render(scene_id, world_kind, W, H, spp, depth, seed, ep, ew, eh, r0, r1, threads,
                                  fb.ctypes.data, st if want_stats else None)
        assert rc == 0
        if want_stats:
            return fb, dict(zip(STAT_NAMES, list(st)))
        return fb

    def scene_dump(self, scene_id, world_kind, W, H, seed=1984):
        kinds = np.zeros(2048, dtype=np.int32)
        boxes = np.zeros((2048, 6), dtype=np.float64)
        cam = np.zeros(27, dtype=np.float64)
        n = self.L.oracle_scene_dump(scene_id, world_kind, W, H, seed, 2048, kinds.ctypes.data, boxes.ctypes.data,
                                     cam.ctypes.data)
        assert n >= 0
        return kinds[:n], boxes[:n], cam

    def rng_stream(self, seed, sequence, n, salt_kind=0):
        raw = np.zeros(n, dtype=np.uint32)
        uni = np.zeros(n, dtype=np.float32)
        self.L.oracle_rng_stream(seed, sequence, salt_kind, n, raw.ctypes.data, uni.ctypes.data)
        return raw, uni

    def rng_state(self, seed, sequence):
        out = np.zeros(6, dtype=np.uint32)
        self.L.oracle_rng_state(seed, sequence, out.ctypes.data)
        return out


class OracleRng:
    """curand_init(seed, sequence, 0) on the oracle side (scene generation: NoiseTexture, random placement)."""

    def __init__(self, seed=1984, sequence=0):
        L = Oracle().L
        L.oracle_rng_new.restype = C.c_void_p
        L.oracle_rng_new.argtypes = [C.c_uint64, C.c_uint64]
        L.oracle_rng_free.argtypes = [C.c_void_p]
        L.oracle_rng_uniform.argtypes = [C.c_void_p]
        L.oracle_rng_uniform.restype = C.c_float
        self.L = L
        self._p = L.oracle_rng_new(seed, sequence)

    def __del__(self):
        if getattr(self, "_p", None):
            self.L.oracle_rng_free(self._p)
            self._p = None

    def uniform(self):
        return self.L.oracle_rng_uniform(self._p)


def _d3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


class OracleScene:
    """The oracle's own object tree built constructor by constructor (oracle_c_* in oracle/rtow_oracle.c).  Method
    names and argument meaning are those of raytracinginoneweekendincuda_amd.Scene (i.e. of the reference's classes), so
    that one scene-building function can be run against both and the frames compared.  Any nesting is allowed."""

    _SIGS = {
        "oracle_c_solid": [C.c_double] * 3, "oracle_c_checker": [C.c_double, C.c_int, C.c_int],
        "oracle_c_image": [C.c_void_p, C.c_int, C.c_int], "oracle_c_noise": [C.c_double, C.c_void_p],
        "oracle_c_lambertian_tex": [C.c_int], "oracle_c_lambertian": [C.c_double] * 3, "oracle_c_metal": [C.c_double] * 4,
        "oracle_c_dielectric": [C.c_double], "oracle_c_light_tex": [C.c_int], "oracle_c_light": [C.c_double] * 3,
        "oracle_c_isotropic_tex": [C.c_int], "oracle_c_isotropic": [C.c_double] * 3,
        "oracle_c_sphere": [C.c_double] * 4 + [C.c_int], "oracle_c_moving_sphere": [C.c_double] * 9 + [C.c_int],
        "oracle_c_quad": [C.c_void_p] * 3 + [C.c_int], "oracle_c_translate": [C.c_int] + [C.c_double] * 3,
        "oracle_c_rotate_y": [C.c_int, C.c_double], "oracle_c_make_box": [C.c_void_p, C.c_void_p, C.c_int],
        "oracle_c_list": [C.c_void_p, C.c_int], "oracle_c_bvh": [C.c_void_p, C.c_int],
        "oracle_c_medium": [C.c_int] + [C.c_double] * 4, "oracle_c_medium_tex": [C.c_int, C.c_double, C.c_int],
        "oracle_c_bounding_box": [C.c_int, C.c_void_p], "oracle_c_set_world": [C.c_int],
        "oracle_c_set_camera": [C.c_void_p] * 3 + [C.c_double] * 6 + [C.c_void_p],
        "oracle_c_render": [C.c_int] * 4 + [C.c_uint64] + [C.c_int] * 3 + [C.c_void_p, C.c_void_p],
    }

    def __init__(self):
        L = Oracle().L
        L.oracle_custom_new.restype = C.c_void_p
        L.oracle_custom_free.argtypes = [C.c_void_p]
        for name, sig in self._SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = [C.c_void_p] + sig
            fn.restype = C.c_int
        self.L = L
        self._p = L.oracle_custom_new()

    def __del__(self):
        if getattr(self, "_p", None):
            self.L.oracle_custom_free(self._p)
            self._p = None

    def _h(self, handle):
        assert handle > 0, "oracle constructor rejected its arguments"
        return handle

    # textures
    def SolidColor(self, c):
        return self._h(self.L.oracle_c_solid(self._p, *map(float, c)))

    def CheckerTexture(self, scale, even, odd):
        return self._h(self.L.oracle_c_checker(self._p, scale, even, odd))

    def ImageTexture(self, rgb):
        if rgb is None:
            return self._h(self.L.oracle_c_image(self._p, None, 0, 0))
        a = np.ascontiguousarray(rgb, dtype=np.uint8)
        return self._h(self.L.oracle_c_image(self._p, a.ctypes.data, a.shape[1], a.shape[0]))

    def NoiseTexture(self, scale, rng):
        return self._h(self.L.oracle_c_noise(self._p, scale, rng._p))

    # materials
    def Lambertian(self, c):
        if isinstance(c, int):
            return self._h(self.L.oracle_c_lambertian_tex(self._p, c))
        return self._h(self.L.oracle_c_lambertian(self._p, *map(float, c)))

    def Metal(self, c, fuzz):
        return self._h(self.L.oracle_c_metal(self._p, float(c[0]), float(c[1]), float(c[2]), fuzz))

    def Dielectric(self, ior):
        return self._h(self.L.oracle_c_dielectric(self._p, ior))

    def DiffuseLight(self, c):
        if isinstance(c, int):
            return self._h(self.L.oracle_c_light_tex(self._p, c))
        return self._h(self.L.oracle_c_light(self._p, *map(float, c)))

    def Isotropic(self, c):
        if isinstance(c, int):
            return self._h(self.L.oracle_c_isotropic_tex(self._p, c))
        return self._h(self.L.oracle_c_isotropic(self._p, *map(float, c)))

    # hittables
    def Sphere(self, center, radius, material):
        return self._h(self.L.oracle_c_sphere(self._p, float(center[0]), float(center[1]), float(center[2]), radius, material))

    def MovingSphere(self, c0, c1, t0, t1, radius, material):
        return self._h(self.L.oracle_c_moving_sphere(self._p, *map(float, c0), *map(float, c1), t0, t1, radius, material))

    def Quad(self, q, u, v, material):
        a, b, c = _d3(q), _d3(u), _d3(v)  # keep the arrays alive across the call
        return self._h(self.L.oracle_c_quad(self._p, C.addressof(a), C.addressof(b), C.addressof(c), material))

    def Translate(self, obj, offset):
        return self._h(self.L.oracle_c_translate(self._p, obj, *map(float, offset)))

    def RotateY(self, obj, degrees):
        return self._h(self.L.oracle_c_rotate_y(self._p, obj, degrees))

    def MakeBox(self, a, b, material):
        lo, hi = _d3(a), _d3(b)
        return self._h(self.L.oracle_c_make_box(self._p, C.addressof(lo), C.addressof(hi), material))

    def HittableList(self, items):
        arr = (C.c_int * max(1, len(items)))(*items)
        return self._h(self.L.oracle_c_list(self._p, C.addressof(arr), len(items)))

    def BvhNode(self, items):
        arr = (C.c_int * max(1, len(items)))(*items)
        root = self._h(self.L.oracle_c_bvh(self._p, C.addressof(arr), len(items)))
        items[:] = list(arr)[: len(items)]
        return root

    def ConstantMedium(self, boundary, density, c):
        if isinstance(c, int):
            return self._h(self.L.oracle_c_medium_tex(self._p, boundary, density, c))
        return self._h(self.L.oracle_c_medium(self._p, boundary, density, *map(float, c)))

    def BoundingBox(self, obj):
        out = (C.c_double * 6)()
        assert self.L.oracle_c_bounding_box(self._p, obj, C.addressof(out)) == 0
        return list(out)

    def SetWorld(self, world):
        assert self.L.oracle_c_set_world(self._p, world) == 0

    def Camera(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist, time0=0.0, time1=0.0,
               background=(0.70, 0.80, 1.00)):
        a, b, c, d = _d3(lookfrom), _d3(lookat), _d3(vup), _d3(background)
        assert self.L.oracle_c_set_camera(self._p, C.addressof(a), C.addressof(b), C.addressof(c), vfov, aspect, aperture,
                                          focus_dist, time0, time1, C.addressof(d)) == 0

    def Commit(self):
        pass

    def render(self, W, H, spp, depth=50, seed=1984, rows=None, threads=0, want_stats=False):
        fb = np.zeros((H, W, 3), dtype=np.float64)
        st = (C.c_uint64 * N_STATS)()
        r0, r1 = rows if rows else (0, H)
        rc = self.L.oracle_c_render(self._p, W, H, spp, depth, seed, r0, r1, threads, fb.ctypes.data,
                                    C.addressof(st) if want_stats else None)
        assert rc == 0
        if want_stats:
            return fb, dict(zip(STAT_NAMES, list(st)))
        return fb


def build_both(build):
    """Run one scene-building function against the product's construction API and against the oracle's.
    `build(s, Rng)` uses only the shared method names; returns (product scene, oracle scene)."""
    import raytracinginoneweekendincuda_amd as rt
    prod, orc = rt.Scene(), OracleScene()
    build(prod, rt.Rng)
    build(orc, OracleRng)
    return prod, orc


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(autouse=True, scope="session")
def one_lane_per_ray_unless_asked():
    """rt_render_params.pixels_per_wave = 0 lets the library give every ray several lanes when a launch has fewer pixels than
    the device has lanes -- which is every frame of a few thousand pixels these tests render.  The tests of the individual
    kernels mean the one-lane-per-ray instantiations, so inside the test session the wrappers' default is 64; the tests of the
    lanes-per-ray scans pass pixels_per_wave themselves (0 = the library's own choice)."""
    import raytracinginoneweekendincuda_amd as rt
    orig_render, orig_params = rt.Scene.render, rt.Film.params

    def render(self, *a, pixels_per_wave=64, **kw):
        return orig_render(self, *a, pixels_per_wave=pixels_per_wave, **kw)

    def params(self, *a, pixels_per_wave=64, **kw):
        return orig_params(self, *a, pixels_per_wave=pixels_per_wave, **kw)

    rt.Scene.render, rt.Film.params = render, params
    yield
    rt.Scene.render, rt.Film.params = orig_render, orig_params


def synthetic_earth(seed=1984, w=256, h=128):
    """Procedural stand-in for earthmap.jpg (the asset cannot travel): smooth bands + noise."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([
        127 + 120 * np.sin(x / w * 6.28 * 3) * np.cos(y / h * 3.14),
        127 + 120 * np.cos(x / w * 6.28 * 2 + 1.0),
        127 + 120 * np.sin(y / h * 6.28 * 2),
    ], axis=-1)
    img = img + rng.integers(-7, 8, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def stb_earth():
    """The bytes the reference's RtwImage::Load hands to ImageTexture for its earthmap.jpg (1024x512 RGB), produced by the
    reference's own stb_image build (tests/golden/make_earth_golden.py; the JPEG itself cannot travel)."""
    with np.load(os.path.join(ROOT, "tests", "golden", "earthmap_stb.npz")) as g:
        return np.ascontiguousarray(g["bytes"])


@pytest.fixture(scope="session")
def earth():
    return stb_earth()


@pytest.fixture(scope="session")
def fake_earth():
    return synthetic_earth()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
