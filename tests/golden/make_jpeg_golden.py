#!/usr/bin/env python3
"""Generates tests/golden/jpeg_cases.npz: small JPEG files (made HERE with Pillow from a synthetic picture: none of them is
reference data) together with the 8-bit pixels the reference's own stb_image build decodes from them (stbi_load through
oracle/_ref/libstb_ref.so = R/StbImageImpl.cpp + R/external/stb_image.h compiled as they lie, `make -C oracle ref`).

The cases cover what the product's restatement of that decoder (csrc/jpeg_decode.cpp) has to get right beyond the
reference's one texture file (1024x512, 4:4:4, no restart interval): the chroma sampling layouts Pillow writes (4:4:4, 4:2:2,
4:2:0 -- stb's horizontal and two-dimensional upsampling filters; its vertical-only and nearest-neighbour ones are restated
but no case here reaches them), greyscale, sizes that are not multiples of the MCU down to 1x1, restart intervals, low
quality (large coefficients: clamping in the inverse DCT), optimised Huffman tables.  Runs only where /root/reference exists.
"""
import ctypes as C
import io
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so")


def picture(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(x / 7.0) * np.cos(y / 11.0), 127 + 120 * np.cos((x + 2 * y) / 13.0),
                    127 + 120 * np.sin((x * y) / 97.0)], axis=-1)
    img[h // 3: h // 2, w // 4: w // 2] = (255, 0, 0)        # hard edges: ringing, clamping
    img[: h // 5, -w // 3:] = (0, 0, 255)
    img += rng.integers(-25, 26, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [  # name, (w, h), Pillow save options
    ("444", (64, 48), dict(quality=90, subsampling=0)),
    ("422", (67, 45), dict(quality=85, subsampling=1)),
    ("420", (70, 53), dict(quality=75, subsampling=2)),
    ("420_tiny", (1, 1), dict(quality=75, subsampling=2)),
    ("420_narrow", (2, 37), dict(quality=75, subsampling=2)),
    ("grey", (50, 33), dict(quality=80)),
    ("420_restart", (96, 64), dict(quality=70, subsampling=2, restart_marker_blocks=3)),
    ("444_restart_rows", (40, 40), dict(quality=95, subsampling=0, restart_marker_rows=1)),
    ("low_quality", (64, 64), dict(quality=5, subsampling=2)),
    ("optimised", (80, 56), dict(quality=60, subsampling=1, optimize=True)),
]


def stb_decode(L, data):
    w, h, ch = C.c_int(), C.c_int(), C.c_int()
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    p = L.stbi_load_from_memory(buf, len(data), C.byref(w), C.byref(h), C.byref(ch), 3)
    assert p, "stbi_load_from_memory failed"
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    L.stbi_image_free(p)
    return out


def main():
    if not os.path.exists(LIB):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    L = C.CDLL(LIB)
    L.stbi_load_from_memory.restype = C.POINTER(C.c_ubyte)
    L.stbi_load_from_memory.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.stbi_image_free.argtypes = [C.c_void_p]
    out = {}
    for k, (name, (w, h), opts) in enumerate(CASES):
        img = picture(w, h, k)
        im = Image.fromarray(img[..., 0] if name == "grey" else img)
        f = io.BytesIO()
        im.save(f, "JPEG", **opts)
        data = f.getvalue()
        out["jpeg_" + name] = np.frombuffer(data, dtype=np.uint8)
        out["rgb_" + name] = stb_decode(L, data)
        print(name, (w, h), len(data), "bytes")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"), **out)


if __name__ == "__main__":
    main()
