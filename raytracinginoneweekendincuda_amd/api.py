"""Host-side mirror of the reference's construction and render interface, over the C-ABI.

Names follow the reference's classes (R/ = reference RayTracinginOneWeekend/): ``Scene.Sphere(center,
radius, material)`` is ``new Sphere(center, radius, material)`` (R/Sphere.h:12), ``Scene.Lambertian``
is R/Material.h:57/63, ``Scene.BvhNode(list)`` is ``new BvhNode(list, 0, n, ...)`` (R/BvhNode.h:50) and
so on; ``Film.render`` is the RenderInit + Render launch pair (R/kernel.cu:675-691) and ``write_ppm`` the
writer at R/kernel.cu:696-721.  Errors surface as RtowError carrying the library's message.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RenderParams, RenderStats, SceneInfo  # noqa: F401


class RtowError(RuntimeError):
    pass


# rt_render_params.flags (include/rtow.h)
FLAG_KEEP_RNG_STATE = 1     # continue the film's saved per-pixel RNG streams (progressive rendering)
FLAG_FORCE_GENERAL = 2      # tests: general kernel even where a specialised instantiation applies
FLAG_OVERDUE_PRIORITY = 4   # diagnostics
FLAG_ACCUMULATE = 8         # with KEEP_RNG_STATE: add this launch's samples to the film's running sums
FLAG_ROW_MAJOR_TILES = 16   # BVH worlds: keep the pixel queue in row-major tile order (no cost ranking)
FLAG_ALWAYS_WALK = 32       # small BVH worlds: walk the tree instead of scanning all leaves
FLAG_ACCELERATE_LISTS = 512  # list worlds of primitives: render through the library's tree (default: scan the list as the reference does)
FLAG_EXACT_SCAN = 256       # sphere-list worlds: the reference's discriminant for every sphere (default: conservative filter first)
FLAG_REFERENCE_TREE = 128   # primitive BVH worlds: walk the reference's own tree (default: the library's SAH tree)
FLAG_FILTER_FP64 = 2048     # sphere-list worlds: the fp64 filter instead of its packed fp32 form (tests, timing)
FLAG_COOP_SINGLE = 1024     # tests: sphere-list worlds, thin waves scan one ray at a time (the older scheme)
FLAG_NO_PIXEL_CLASSES = 64  # sphere-list worlds: one launch for all pixels (no separate launch for the long-chain pixels)


# rt_scene_set_options (read by the next commit)
SCENE_PLAIN_QUADS = 1           # every quad takes the general test; boxes stay lists of six quads
SCENE_REFERENCE_TREE_ONLY = 2   # no library tree for primitive worlds


def lib():
    return _lib.load()


def library_path():
    return _lib.LIB_PATH


def _err():
    return lib().rt_last_error().decode()


def _check(status):
    if status != 0:
        raise RtowError(f"status {status}: {_err()}")


def _h(handle):
    if not handle:
        raise RtowError(_err())
    return handle


def _v3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


class Rng:
    """curand_init(seed, sequence, 0) + curand_uniform (R/kernel.cu:101-107, RND at :157)."""

    def __init__(self, seed=1984, sequence=0, salt_kind=0):
        self._p = lib().rt_rng_create_salted(seed, sequence, salt_kind)

    def __del__(self):
        if getattr(self, "_p", None):
            lib().rt_rng_destroy(self._p)
            self._p = None

    def uniform(self):
        return lib().rt_rng_uniform(self._p)

    def next_u32(self):
        return lib().rt_rng_next_u32(self._p)

    def state(self):
        out = (C.c_uint32 * 6)()
        lib().rt_rng_state(self._p, out)
        return list(out)


class Scene:
    def __init__(self):
        self._p = lib().rt_scene_create()
        self._keep = []

    def __del__(self):
        if getattr(self, "_p", None):
            lib().rt_scene_destroy(self._p)
            self._p = None

    def set_options(self, options):
        _check(lib().rt_scene_set_options(self._p, options))

    # ---- textures (R/Texture.h) ----
    def SolidColor(self, c):
        return _h(lib().rt_solid_color(self._p, *map(float, c)))

    def CheckerTexture(self, scale, even, odd):
        return _h(lib().rt_checker_texture(self._p, scale, even, odd))

    def ImageTexture(self, rgb):
        if rgb is None:
            return _h(lib().rt_image_texture(self._p, None, 0, 0))
        a = np.ascontiguousarray(rgb, dtype=np.uint8)
        return _h(lib().rt_image_texture(self._p, a.ctypes.data, a.shape[1], a.shape[0]))

    def NoiseTexture(self, scale, rng):
        return _h(lib().rt_noise_texture(self._p, scale, rng._p))

    # ---- materials ----
    def Lambertian(self, c):
        if isinstance(c, int):
            return _h(lib().rt_lambertian_tex(self._p, c))
        return _h(lib().rt_lambertian(self._p, *map(float, c)))

    def Metal(self, c, fuzz):
        return _h(lib().rt_metal(self._p, float(c[0]), float(c[1]), float(c[2]), fuzz))

    def Dielectric(self, ior):
        return _h(lib().rt_dielectric(self._p, ior))

    def DiffuseLight(self, c):
        if isinstance(c, int):
            return _h(lib().rt_diffuse_light_tex(self._p, c))
        return _h(lib().rt_diffuse_light(self._p, *map(float, c)))

    def Isotropic(self, c):
        if isinstance(c, int):
            return _h(lib().rt_isotropic_tex(self._p, c))
        return _h(lib().rt_isotropic(self._p, *map(float, c)))

    # ---- hittables ----
    def Sphere(self, center, radius, material):
        return _h(lib().rt_sphere(self._p, float(center[0]), float(center[1]), float(center[2]), radius, material))

    def MovingSphere(self, c0, c1, t0, t1, radius, material):
        return _h(lib().rt_moving_sphere(self._p, *map(float, c0), *map(float, c1), t0, t1, radius, material))

    def Quad(self, q, u, v, material):
        return _h(lib().rt_quad(self._p, _v3(q), _v3(u), _v3(v), material))

    def Translate(self, obj, offset):
        return _h(lib().rt_translate(self._p, obj, *map(float, offset)))

    def RotateY(self, obj, degrees):
        return _h(lib().rt_rotate_y(self._p, obj, degrees))

    def MakeBox(self, a, b, material):
        return _h(lib().rt_make_box(self._p, _v3(a), _v3(b), material))

    def HittableList(self, items):
        arr = (C.c_uint32 * max(1, len(items)))(*items)
        return _h(lib().rt_hittable_list(self._p, arr, len(items)))

    def ConstantMedium(self, boundary, density, c):
        if isinstance(c, int):
            return _h(lib().rt_constant_medium_tex(self._p, boundary, density, c))
        return _h(lib().rt_constant_medium(self._p, boundary, density, *map(float, c)))

    def BvhNode(self, items):
        """Sorts ``items`` in place like the reference sorts list[] (R/BvhNode.h:180-193)."""
        arr = (C.c_uint32 * max(1, len(items)))(*items)
        root = _h(lib().rt_bvh_node(self._p, arr, len(items)))
        items[:] = list(arr)[: len(items)]
        return root

    def BoundingBox(self, obj):
        out = (C.c_double * 6)()
        _check(lib().rt_hittable_bounding_box(self._p, obj, out))
        return list(out)

    # ---- world / camera / commit ----
    def SetWorld(self, world):
        _check(lib().rt_scene_set_world(self._p, world))

    def Camera(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist, time0=0.0, time1=0.0,
               background=(0.70, 0.80, 1.00)):
        _check(lib().rt_scene_set_camera(self._p, _v3(lookfrom), _v3(lookat), _v3(vup), vfov, aspect, aperture,
                                         focus_dist, time0, time1, _v3(background)))

    def Commit(self):
        _check(lib().rt_scene_commit(self._p))

    def build_builtin(self, scene_id, world_kind, width, height, seed=1984, earth=None):
        if earth is not None:
            earth = np.ascontiguousarray(earth, dtype=np.uint8)
            self._keep.append(earth)
            _check(lib().rt_scene_build_builtin(self._p, scene_id, world_kind, width, height, seed, earth.ctypes.data,
                                                earth.shape[1], earth.shape[0]))
        else:
            _check(lib().rt_scene_build_builtin(self._p, scene_id, world_kind, width, height, seed, None, 0, 0))
        return self

    # ---- introspection ----
    def info(self):
        out = SceneInfo()
        _check(lib().rt_scene_get_info(self._p, C.byref(out)))
        return {n: getattr(out, n) for n, _ in SceneInfo._fields_ if n != "reserved"}

    def dump_leaves(self):
        n = lib().rt_scene_dump_leaves(self._p, 0, None, None)
        if n < 0:
            raise RtowError(_err())
        kinds = np.zeros(max(n, 1), dtype=np.int32)
        boxes = np.zeros((max(n, 1), 6), dtype=np.float64)
        lib().rt_scene_dump_leaves(self._p, n, kinds.ctypes.data_as(C.POINTER(C.c_int)), boxes.ctypes.data_as(_lib.D3))
        return kinds[:n], boxes[:n]

    def dump_nodes(self):
        n = lib().rt_scene_dump_nodes(self._p, 0, None, None)
        if n < 0:
            raise RtowError(_err())
        boxes = np.zeros((max(n, 1), 6), dtype=np.float64)
        abe = np.zeros((max(n, 1), 3), dtype=np.uint32)
        lib().rt_scene_dump_nodes(self._p, n, boxes.ctypes.data_as(_lib.D3), abe.ctypes.data_as(C.POINTER(C.c_uint32)))
        return boxes[:n], abe[:n]

    def dump_fast_nodes(self):
        """The library's own tree for a primitive-only BVH world: boxes (n, 6), leaf refs (n, 2), octant links (n, 8, 2)."""
        n = lib().rt_scene_dump_fast_nodes(self._p, 0, None, None, None)
        if n < 0:
            raise RtowError(_err())
        boxes = np.zeros((max(n, 1), 6), dtype=np.float64)
        ab = np.zeros((max(n, 1), 2), dtype=np.uint32)
        links = np.zeros((max(n, 1), 8, 2), dtype=np.uint16)
        lib().rt_scene_dump_fast_nodes(self._p, n, boxes.ctypes.data_as(_lib.D3), ab.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       links.ctypes.data_as(C.POINTER(C.c_uint16)))
        return boxes[:n], ab[:n], links[:n]

    def dump_camera(self):
        out = np.zeros(27, dtype=np.float64)
        _check(lib().rt_scene_dump_camera(self._p, out.ctypes.data_as(_lib.D3)))
        return out

    def upload(self, device=0):
        _check(lib().rt_scene_upload(self._p, device))

    # ---- one-call render on one GPU ----
    def render(self, width, height, spp, max_depth=50, seed=1984, variant=0, device=0, flags=0, coop_threshold=0,
               overdue=0, shade_batch=0, max_blocks_per_cu=0, pixels_per_wave=0):
        p = RenderParams(width, height, spp, max_depth, seed, 8, 0, 1, variant, device, flags, None, coop_threshold, overdue,
                         shade_batch, max_blocks_per_cu, pixels_per_wave, 0)
        frame = np.zeros((height, width, 3), dtype=np.float64)
        st = RenderStats()
        _check(lib().rt_render(self._p, C.byref(p), frame.ctypes.data_as(_lib.D3), C.byref(st)))
        return frame, st


def builtin_scene(scene_id, world_kind, width, height, seed=1984, earth=None):
    """CreateWorld(sceneId) (R/kernel.cu:176-543); world_kind 0 = BvhNode world, 1 = HittableList world."""
    return Scene().build_builtin(scene_id, world_kind, width, height, seed, earth)


class Film:
    """frameBuffer + randState of one GPU (R/kernel.cu:606-613), restricted to this rank's row stripes."""

    def __init__(self, width, height, device=0, stripe_rows=8, rank=0, world_size=1):
        self.width, self.height, self.device = width, height, device
        self.stripe_rows, self.rank, self.world_size = stripe_rows, rank, world_size
        self._p = lib().rt_film_create(device, width, height, stripe_rows, rank, world_size)
        if not self._p:
            raise RtowError(_err())

    def __del__(self):
        if getattr(self, "_p", None):
            lib().rt_film_destroy(self._p)   # waits for a launch still in flight
            self._p = None
        self._scene = None

    def params(self, spp, max_depth=50, seed=1984, variant=0, flags=0, stream=None, coop_threshold=0, overdue=0,
               shade_batch=0, max_blocks_per_cu=0, pixels_per_wave=0):
        return RenderParams(self.width, self.height, spp, max_depth, seed, self.stripe_rows, self.rank, self.world_size,
                            variant, self.device, flags, stream, coop_threshold, overdue, shade_batch, max_blocks_per_cu,
                            pixels_per_wave, 0)

    def launch(self, scene, params):
        _check(lib().rt_render_launch(scene._p, self._p, C.byref(params)))
        self._scene = scene   # the kernel reads the scene's tables: keep it alive until finish()

    def finish(self, scene=None):
        st = RenderStats()
        scene = scene if scene is not None else getattr(self, "_scene", None)
        try:
            _check(lib().rt_render_finish(scene._p if scene is not None else None, self._p, C.byref(st)))
        finally:
            self._scene = None
        return st

    def render(self, scene, spp, **kw):
        self.launch(scene, self.params(spp, **kw))
        return self.finish(scene)

    def device_pixels(self):
        return lib().rt_film_device_pixels(self._p), lib().rt_film_pixel_bytes(self._p)

    def bind_pixels(self, device_ptr):
        """Render into caller-owned device memory (a torch tensor's data_ptr()) of pixel_bytes bytes."""
        _check(lib().rt_film_bind_pixels(self._p, device_ptr))

    @property
    def pixel_bytes(self):
        return lib().rt_film_pixel_bytes(self._p)

    def download(self):
        frame = np.zeros((self.height, self.width, 3), dtype=np.float64)
        _check(lib().rt_film_download(self._p, frame.ctypes.data_as(_lib.D3), self.width, self.height))
        return frame


def rtwimage_bytes(decoded_rgb):
    """RtwImage::Load's pixel conversion (stb linearisation with gamma 2.2, then FloatToByte): takes decoded 8-bit
    sRGB pixels (H, W, 3) from any JPEG decoder and returns the bytes the reference hands to ImageTexture."""
    a = np.ascontiguousarray(decoded_rgb, dtype=np.uint8)
    out = np.empty_like(a)
    lib().rt_rtwimage_bytes(a.ctypes.data, a.size, out.ctypes.data)
    return out


def _take_image(ptr, w, h):
    try:
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_ubyte)), shape=(h.value, w.value, 3)).copy()
    finally:
        lib().rt_image_free(ptr)


def jpeg_decode(data):
    """The 8-bit sRGB pixels (H, W, 3) of a sequential Huffman JPEG exactly as the reference's stb_image decodes them
    (csrc/jpeg_decode.cpp); raises RtowError for what that restatement does not decode (progressive, CMYK, ...)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    ptr, w, h = C.c_void_p(), C.c_int(), C.c_int()
    _check(lib().rt_jpeg_decode(buf.ctypes.data, buf.size, C.byref(ptr), C.byref(w), C.byref(h)))
    return _take_image(ptr, w, h)


def load_image(path):
    """RtwImage::Load (R/RtwImage.h:51-87): the bytes the reference hands to ImageTexture for a JPEG file, bit for bit.
    Returns None if the file cannot be read or decoded, which ImageTexture turns into the reference's cyan fallback."""
    ptr, w, h = C.c_void_p(), C.c_int(), C.c_int()
    if lib().rt_rtwimage_load(str(path).encode(), C.byref(ptr), C.byref(w), C.byref(h)) != 0:
        return None
    return _take_image(ptr, w, h)


def stripe_rows(height, stripe, rank, world_size):
    n = lib().rt_stripe_rows(height, stripe, rank, world_size, None, 0)
    if n < 0:
        raise RtowError("bad stripe arguments")
    rows = (C.c_int * max(n, 1))()
    lib().rt_stripe_rows(height, stripe, rank, world_size, rows, n)
    return list(rows)[:n]


def deinterleave(gathered, width, height, stripe, world_size):
    """gathered: (world_size, rows_max*width*3) float64, as gathered rank-major over RCCL."""
    g = np.ascontiguousarray(gathered, dtype=np.float64)
    frame = np.zeros((height, width, 3), dtype=np.float64)
    _check(lib().rt_deinterleave(g.ctypes.data_as(_lib.D3), width, height, stripe, world_size, g.shape[1],
                                 frame.ctypes.data_as(_lib.D3)))
    return frame


def write_ppm_binary(path, frame):
    f = np.ascontiguousarray(frame, dtype=np.float64)
    _check(lib().rt_write_ppm_binary(str(path).encode(), f.ctypes.data_as(_lib.D3), f.shape[1], f.shape[0]))


def write_pfm(path, frame):
    f = np.ascontiguousarray(frame, dtype=np.float64)
    _check(lib().rt_write_pfm(str(path).encode(), f.ctypes.data_as(_lib.D3), f.shape[1], f.shape[0]))


def write_ppm(path, frame):
    f = np.ascontiguousarray(frame, dtype=np.float64)
    _check(lib().rt_write_ppm(str(path).encode(), f.ctypes.data_as(_lib.D3), f.shape[1], f.shape[0]))
