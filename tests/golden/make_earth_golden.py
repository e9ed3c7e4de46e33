#!/usr/bin/env python3
"""Generates tests/golden/earthmap_stb.npz: the bytes the reference hands to ImageTexture for earthmap.jpg.

Runs only where /root/reference exists (the build container).  Uses oracle/_ref/libstb_ref.so, which is the
reference's own R/StbImageImpl.cpp + R/external/stb_image.h compiled as they lie (oracle/Makefile, target `ref`):

  stbi_loadf("earthmap.jpg", 3 channels)      R/RtwImage.h:54   (JPEG decode + stbi__ldr_to_hdr, stb_image.h:1858-1879)
  FloatToByte on every float                  R/RtwImage.h:66-67,100-105  (restated below in float32: the class itself
                                              cannot be compiled here, it includes cuda_runtime.h)

Stored: `bytes` (H, W, 3) uint8 = ImageTexture's data for the whole image; `srgb_crop`/`float_crop` = stb's 8-bit decode
(stbi_load) and its float output for one crop, to pin rt_rtwimage_bytes' restatement of stbi__ldr_to_hdr; and how far
Pillow's (libjpeg) decode of the same file is from stb's, as documentation of what a different decoder costs.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
JPG = "/root/reference/earthmap.jpg"
LIB = os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so")
CROP = (slice(192, 256), slice(448, 576))  # 64 x 128 pixels around the middle


def main():
    if not os.path.exists(LIB):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    L = C.CDLL(LIB)
    L.stbi_loadf.restype = C.POINTER(C.c_float)
    L.stbi_loadf.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.stbi_load.restype = C.POINTER(C.c_ubyte)
    L.stbi_load.argtypes = L.stbi_loadf.argtypes
    L.stbi_image_free.argtypes = [C.c_void_p]
    w, h, ch = C.c_int(), C.c_int(), C.c_int()
    fp = L.stbi_loadf(JPG.encode(), C.byref(w), C.byref(h), C.byref(ch), 3)
    assert fp, "stbi_loadf failed"
    W, H = w.value, h.value
    f = np.ctypeslib.as_array(fp, shape=(H, W, 3)).copy()
    L.stbi_image_free(fp)
    bp = L.stbi_load(JPG.encode(), C.byref(w), C.byref(h), C.byref(ch), 3)
    srgb = np.ctypeslib.as_array(bp, shape=(H, W, 3)).copy()
    L.stbi_image_free(bp)

    # RtwImage::FloatToByte, float32 arithmetic, truncating cast
    f32 = f.astype(np.float32)
    scaled = (np.float32(256.0) * f32).astype(np.float32)
    out = np.where(f32 <= 0, 0, np.where(f32 >= 1, 255, np.minimum(scaled, 255.99).astype(np.uint8))).astype(np.uint8)

    note = {}
    try:
        from PIL import Image
        pil = np.asarray(Image.open(JPG).convert("RGB"))
        d = np.abs(pil.astype(int) - srgb.astype(int))
        note = {"pillow_vs_stb_max_abs": int(d.max()), "pillow_vs_stb_frac_bytes_differing": float((d > 0).mean())}
    except Exception as e:  # noqa: BLE001
        note = {"pillow": f"unavailable: {e}"}
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "earthmap_stb.npz"), bytes=out, srgb_crop=srgb[CROP],
                        float_crop=f[CROP], crop=np.array([CROP[0].start, CROP[0].stop, CROP[1].start, CROP[1].stop]),
                        channels_in_file=np.array(ch.value), **{k: np.array(v) for k, v in note.items()})
    print(f"earthmap.jpg {W}x{H}x{ch.value}: {out.size} bytes written; {note}")


if __name__ == "__main__":
    main()
