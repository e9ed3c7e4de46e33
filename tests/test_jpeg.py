"""Image ingest: csrc/jpeg_decode.cpp restates what the reference's stb_image does to a sequential Huffman JPEG;
rt_rtwimage_load is RtwImage::Load (R/RtwImage.h:51-87).  Pinned by the reference's own decoder: tests/golden/jpeg_cases.npz
holds JPEG files made with Pillow (tests/golden/make_jpeg_golden.py) and the pixels R/StbImageImpl.cpp decodes from them;
where the reference checkout is present, its earthmap.jpg itself must come out as tests/golden/earthmap_stb.npz, which the
reference's stb build produced.  No GPU needed."""
import os

import numpy as np
import pytest

import raytracinginoneweekendincuda_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"))
NAMES = sorted(k[5:] for k in CASES.files if k.startswith("jpeg_"))


@pytest.mark.parametrize("name", NAMES)
def test_decoder_gives_the_reference_decoders_pixels(name):
    got = rt.jpeg_decode(CASES["jpeg_" + name].tobytes())
    want = CASES["rgb_" + name]
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{name}: {(got != want).sum()} bytes differ, max {np.abs(got.astype(int) - want.astype(int)).max()}"


def test_cases_cover_the_sampling_layouts():
    assert {"444", "422", "420", "grey", "420_restart", "low_quality"} <= set(NAMES)


@pytest.mark.skipif(not os.path.exists("/root/reference/earthmap.jpg"), reason="the reference's texture file is only in the build container")
def test_rtwimage_load_reproduces_the_references_earth_texture():
    got = rt.load_image("/root/reference/earthmap.jpg")
    with np.load(os.path.join(ROOT, "tests", "golden", "earthmap_stb.npz")) as g:
        want = g["bytes"]
    assert got is not None and got.shape == want.shape == (512, 1024, 3)
    assert np.array_equal(got, want)


def test_what_is_not_decoded_is_an_error_not_a_wrong_picture(tmp_path):
    from PIL import Image
    import io
    f = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), dtype=np.uint8)).save(f, "JPEG", progressive=True)
    with pytest.raises(rt.RtowError, match="progressive"):
        rt.jpeg_decode(f.getvalue())
    with pytest.raises(rt.RtowError):
        rt.jpeg_decode(b"not a jpeg at all")
    with pytest.raises(rt.RtowError):
        rt.jpeg_decode(CASES["jpeg_444"].tobytes()[:40])          # truncated inside the headers
    assert rt.load_image(str(tmp_path / "missing.jpg")) is None   # RtwImage::Load's failure: ImageTexture(None) renders cyan
    cut = rt.jpeg_decode(CASES["jpeg_420"].tobytes()[:-200])       # truncated scan data: decoded as far as it goes
    assert cut.shape == CASES["rgb_420"].shape
