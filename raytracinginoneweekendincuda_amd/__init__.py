"""MI355X-native path tracer: host-side mirror of the reference's construction/render API.

The compute path is the hand-written gfx950 HIP library ``librtow_hip.so`` (C-ABI in
``include/rtow.h``).  There is no CPU fallback: importing works anywhere (so that host-side scene
logic can be tested), rendering raises unless the HIP library is loaded and a GPU is present.
"""
from .api import (  # noqa: F401
    RtowError,
    Rng,
    Scene,
    Film,
    RenderParams,
    RenderStats,
    builtin_scene,
    stripe_rows,
    deinterleave,
    write_ppm,
    write_ppm_binary,
    write_pfm,
    rtwimage_bytes,
    load_image,
    jpeg_decode,
    library_path,
    lib,
    FLAG_KEEP_RNG_STATE,
    FLAG_FORCE_GENERAL,
    FLAG_OVERDUE_PRIORITY,
    FLAG_ACCUMULATE,
    FLAG_ROW_MAJOR_TILES,
    FLAG_ALWAYS_WALK, FLAG_NO_PIXEL_CLASSES, FLAG_REFERENCE_TREE, FLAG_EXACT_SCAN, FLAG_ACCELERATE_LISTS, FLAG_COOP_SINGLE, FLAG_FILTER_FP64,
    SCENE_PLAIN_QUADS, SCENE_REFERENCE_TREE_ONLY,
)

__all__ = [
    "RtowError", "Rng", "Scene", "Film", "RenderParams", "RenderStats", "builtin_scene",
    "stripe_rows", "deinterleave", "write_ppm", "write_ppm_binary", "write_pfm", "rtwimage_bytes", "load_image", "library_path", "lib",
    "FLAG_KEEP_RNG_STATE", "FLAG_FORCE_GENERAL", "FLAG_OVERDUE_PRIORITY", "FLAG_ACCUMULATE", "FLAG_ROW_MAJOR_TILES", "FLAG_ALWAYS_WALK",
]
