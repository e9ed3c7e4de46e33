"""the image texture's bytes with stb semantics.

tests/golden/earthmap_stb.npz was produced by the reference's OWN stb_image translation unit (R/StbImageImpl.cpp +
R/external/stb_image.h compiled as they lie into oracle/_ref/libstb_ref.so, see tests/golden/make_earth_golden.py):
`bytes` is exactly what RtwImage::Load (R/RtwImage.h:51-87) hands to ImageTexture for the reference's earthmap.jpg.
"""
import ctypes as C
import os

import numpy as np
import pytest

import raytracinginoneweekendincuda_amd as rt
from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "earthmap_stb.npz")


def test_rtwimage_bytes_reproduces_stb_ldr_to_hdr_and_float_to_byte():
    """rt_rtwimage_bytes = FloatToByte(stbi__ldr_to_hdr(decoded byte)) for every byte value that occurs: stb's own 8-bit
    decode of a crop, pushed through the product's conversion, equals the bytes stb's float path + FloatToByte gave."""
    g = np.load(GOLD)
    r0, r1, c0, c1 = (int(x) for x in g["crop"])
    got = rt.rtwimage_bytes(g["srgb_crop"])
    assert np.array_equal(got, g["bytes"][r0:r1, c0:c1])
    # and the float stage on its own: (float)pow(b / 255.0f, 2.2f) for all 256 inputs, against stb's output
    b = g["srgb_crop"].astype(np.float32) / np.float32(255.0)
    want = np.power(b.astype(np.float64), np.float64(np.float32(2.2))).astype(np.float32)
    assert np.array_equal(want.view(np.uint32), g["float_crop"].view(np.uint32))
    assert len(np.unique(g["srgb_crop"])) > 200, "the crop should exercise most byte values"


def test_golden_image_shape_and_documented_decoder_gap():
    g = np.load(GOLD)
    assert g["bytes"].shape == (512, 1024, 3) and g["bytes"].dtype == np.uint8 and int(g["channels_in_file"]) == 3
    # Pillow (libjpeg) vs stb on the same file, recorded when the fixture was made: what using another decoder costs
    assert int(g["pillow_vs_stb_max_abs"]) <= 3 and float(g["pillow_vs_stb_frac_bytes_differing"]) < 0.01


@pytest.mark.skipif(not (os.path.exists("/root/reference/earthmap.jpg") and
                         os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so"))),
                    reason="needs the reference checkout and oracle/_ref (build container only)")
def test_fixture_is_what_the_reference_stb_build_decodes_today():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so"))
    L.stbi_load.restype = C.POINTER(C.c_ubyte)
    L.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.stbi_image_free.argtypes = [C.c_void_p]
    w, h, ch = C.c_int(), C.c_int(), C.c_int()
    p = L.stbi_load(b"/root/reference/earthmap.jpg", C.byref(w), C.byref(h), C.byref(ch), 3)
    assert p
    srgb = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    L.stbi_image_free(p)
    assert np.array_equal(rt.rtwimage_bytes(srgb), np.load(GOLD)["bytes"])
