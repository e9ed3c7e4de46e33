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


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


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


@pytest.fixture(scope="session")
def earth():
    return synthetic_earth()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
