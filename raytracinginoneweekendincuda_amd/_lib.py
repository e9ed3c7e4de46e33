"""ctypes binding of include/rtow.h.  Fails loudly when the HIP library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTOW_LIB_PATH") or os.path.join(_HERE, "librtow_hip.so")  # override: A/B builds only


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("samples_per_pixel", C.c_int32), ("max_depth", C.c_int32),
        ("seed", C.c_uint64), ("stripe_rows", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
        ("variant", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32), ("stream", C.c_void_p),
        ("coop_threshold", C.c_int32), ("overdue_rays_per_sample", C.c_int32),
        ("shade_batch", C.c_int32), ("max_blocks_per_cu", C.c_int32), ("pixels_per_wave", C.c_int32), ("reserved0", C.c_int32),
    ]


class RenderStats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64), ("rays", C.c_uint64), ("seconds_seed", C.c_double), ("seconds_render", C.c_double),
        ("pixels", C.c_uint32), ("rows", C.c_uint32), ("kernel_vgprs", C.c_uint32), ("lds_bytes", C.c_uint32),
        ("kernel_kind", C.c_uint32), ("pixels_per_wave", C.c_uint32),
    ]


class SceneInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "world_kind", "n_leaves", "n_nodes", "n_spheres", "n_moving_spheres", "n_quads", "n_objects", "n_xforms",
        "n_media", "n_materials", "n_textures", "n_perlin", "n_images", "table_bytes", "image_bytes")] + [
        ("reserved", C.c_uint32 * 3)]


H = C.c_uint32
P = C.c_void_p
D = C.c_double
I = C.c_int
D3 = C.POINTER(C.c_double)

# name -> (restype, argtypes); exactly the symbols include/rtow.h declares
SIGNATURES = {
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
    "rt_rng_create": (P, [C.c_uint64, C.c_uint64]),
    "rt_rng_create_salted": (P, [C.c_uint64, C.c_uint64, I]),
    "rt_rng_destroy": (None, [P]),
    "rt_rng_uniform": (C.c_float, [P]),
    "rt_rng_next_u32": (C.c_uint32, [P]),
    "rt_rng_state": (None, [P, C.POINTER(C.c_uint32)]),
    "rt_scene_create": (P, []),
    "rt_scene_destroy": (None, [P]),
    "rt_scene_set_options": (I, [P, C.c_uint32]),
    "rt_solid_color": (H, [P, D, D, D]),
    "rt_checker_texture": (H, [P, D, H, H]),
    "rt_image_texture": (H, [P, P, I, I]),
    "rt_noise_texture": (H, [P, D, P]),
    "rt_rtwimage_bytes": (None, [P, C.c_size_t, P]),
    "rt_rtwimage_load": (I, [C.c_char_p, C.POINTER(P), C.POINTER(I), C.POINTER(I)]),
    "rt_jpeg_decode": (I, [P, C.c_size_t, C.POINTER(P), C.POINTER(I), C.POINTER(I)]),
    "rt_image_free": (None, [P]),
    "rt_lambertian": (H, [P, D, D, D]),
    "rt_lambertian_tex": (H, [P, H]),
    "rt_metal": (H, [P, D, D, D, D]),
    "rt_dielectric": (H, [P, D]),
    "rt_diffuse_light": (H, [P, D, D, D]),
    "rt_diffuse_light_tex": (H, [P, H]),
    "rt_isotropic": (H, [P, D, D, D]),
    "rt_isotropic_tex": (H, [P, H]),
    "rt_sphere": (H, [P, D, D, D, D, H]),
    "rt_moving_sphere": (H, [P, D, D, D, D, D, D, D, D, D, H]),
    "rt_quad": (H, [P, D3, D3, D3, H]),
    "rt_translate": (H, [P, H, D, D, D]),
    "rt_rotate_y": (H, [P, H, D]),
    "rt_make_box": (H, [P, D3, D3, H]),
    "rt_hittable_list": (H, [P, C.POINTER(H), I]),
    "rt_constant_medium": (H, [P, H, D, D, D, D]),
    "rt_constant_medium_tex": (H, [P, H, D, H]),
    "rt_bvh_node": (H, [P, C.POINTER(H), I]),
    "rt_hittable_bounding_box": (I, [P, H, D3]),
    "rt_scene_set_world": (I, [P, H]),
    "rt_scene_set_camera": (I, [P, D3, D3, D3, D, D, D, D, D, D, D3]),
    "rt_scene_build_builtin": (I, [P, I, I, I, I, C.c_uint64, P, I, I]),
    "rt_scene_commit": (I, [P]),
    "rt_scene_get_info": (I, [P, C.POINTER(SceneInfo)]),
    "rt_scene_dump_leaves": (I, [P, I, C.POINTER(C.c_int), D3]),
    "rt_scene_dump_nodes": (I, [P, I, D3, C.POINTER(C.c_uint32)]),
    "rt_scene_dump_fast_nodes": (I, [P, I, D3, C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)]),
    "rt_scene_dump_camera": (I, [P, D3]),
    "rt_stripe_rows": (I, [I, I, I, I, C.POINTER(C.c_int), I]),
    "rt_film_create": (P, [I, I, I, I, I, I]),
    "rt_film_destroy": (None, [P]),
    "rt_film_device_pixels": (P, [P]),
    "rt_film_pixel_bytes": (C.c_size_t, [P]),
    "rt_film_bind_pixels": (I, [P, P]),
    "rt_scene_upload": (I, [P, I]),
    "rt_render_launch": (I, [P, P, C.POINTER(RenderParams)]),
    "rt_render_finish": (I, [P, P, C.POINTER(RenderStats)]),
    "rt_film_download": (I, [P, D3, I, I]),
    "rt_deinterleave": (I, [D3, I, I, I, I, C.c_size_t, D3]),
    "rt_render": (I, [P, C.POINTER(RenderParams), D3, C.POINTER(RenderStats)]),
    "rt_write_ppm": (I, [C.c_char_p, D3, I, I]),
    "rt_write_ppm_binary": (I, [C.c_char_p, D3, I, I]),
    "rt_write_pfm": (I, [C.c_char_p, D3, I, I]),
}

_lib = None


def load():
    """Load librtow_hip.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the render path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export what rtow.h declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
