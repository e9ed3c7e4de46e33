#!/usr/bin/env python3
"""Generates tests/golden/oracle_golden.npz from the CPU oracle (oracle/liboracle.so).

These fixtures are REGRESSION PINS of the oracle, not reference outputs: the reference ships no golden data
and cannot be built in this image (see DESIGN.md "Oracle"), so parity with the reference stays "unpinned".
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import Oracle, synthetic_earth  # noqa: E402

W, H, SPP = 32, 16, 2


def main():
    orc = Oracle()
    earth = synthetic_earth()
    out = {"earth": earth, "geometry": np.array([W, H, SPP, 50, 1984])}
    for scene in range(12):
        for world in (0, 1):
            fb, st = orc.render(scene, world, W, H, SPP, earth=earth, want_stats=True)
            out[f"frame_s{scene}_w{world}"] = fb
            out[f"rays_s{scene}_w{world}"] = np.array([st["rays"]], dtype=np.uint64)
    for k, (seed, seq) in enumerate([(1984, 0), (1984, 1), (1984, 959999), (1984, 2559999), (7, 2**40 + 3)]):
        raw, uni = orc.rng_stream(seed, seq, 16)
        out[f"rng_{k}_key"] = np.array([seed, seq], dtype=np.uint64)
        out[f"rng_{k}_raw"] = raw
        out[f"rng_{k}_uniform"] = uni
    for scene in range(12):
        kinds, boxes, cam = orc.scene_dump(scene, 0, 1200, 800)
        out[f"leaves_s{scene}_kinds"] = kinds.astype(np.int8)
        out[f"leaves_s{scene}_boxes"] = boxes
        out[f"camera_s{scene}"] = cam
    path = os.path.join(HERE, "oracle_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
