"""XORWOW: product header (rng.h via the C-ABI) == oracle's independent C implementation, and both ==
rocRAND's engine when given rocRAND's seed salts (validates step + 2^67 sequence jump)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import raytracinginoneweekendincuda_amd as rt

ROCRAND_SRC = r"""
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_xorwow.h>
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
  unsigned long long seed = strtoull(argv[1], 0, 10), seq = strtoull(argv[2], 0, 10);
  int n = atoi(argv[3]);
  rocrand_state_xorwow s; rocrand_init(seed, seq, 0ULL, &s);
  for (int i = 0; i < n; i++) printf("%u\n", rocrand(&s));
  return 0;
}
"""

CASES = [(1984, 0), (1984, 1), (1984, 1199 * 800 + 7), (1984, 2559999), (0, 5), (2**63 + 12345, 2**40 + 3),
         (7, 0xFFFFFFFFFFFFFFFF)]


def test_product_rng_equals_oracle_rng(oracle):
    for seed, seq in CASES:
        r = rt.Rng(seed, seq)
        assert list(oracle.rng_state(seed, seq)) == r.state()
        raw, uni = oracle.rng_stream(seed, seq, 64)
        got = np.array([r.uniform() for _ in range(64)], dtype=np.float32)
        assert np.array_equal(got.view(np.uint32), uni.view(np.uint32))
        r2 = rt.Rng(seed, seq)
        assert [r2.next_u32() for _ in range(64)] == list(raw)


def test_uniform_is_half_open_at_zero_and_formula():
    # curand_uniform(x) = x * 2^-32 + 2^-33 in fp32: in (0, 1]
    r = rt.Rng(1984, 0)
    r2 = rt.Rng(1984, 0)
    for _ in range(2000):
        x = r2.next_u32()
        u = np.float32(r.uniform())
        expect = np.float32(np.float32(x) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33))
        assert u == expect and 0.0 < u <= 1.0


def test_sequences_are_disjoint_streams():
    a = [rt.Rng(1984, 0).next_u32() for _ in range(1)]
    b = [rt.Rng(1984, 1).next_u32() for _ in range(1)]
    assert a != b


@pytest.mark.skipif(not os.path.exists("/opt/rocm/include/rocrand/rocrand_xorwow.h") or shutil.which("g++") is None,
                    reason="rocRAND headers or g++ not present")
def test_step_and_sequence_jump_match_rocrand(tmp_path, oracle):
    src = tmp_path / "rr.cpp"
    src.write_text(ROCRAND_SRC)
    exe = tmp_path / "rr"
    subprocess.check_call(["g++", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(src), "-o", str(exe)])
    for seed, seq in CASES:
        out = subprocess.check_output([str(exe), str(seed), str(seq), "32"]).split()
        want = [int(x) for x in out]
        r = rt.Rng(seed, seq, salt_kind=1)
        assert [r.next_u32() for _ in range(32)] == want
        raw, _ = oracle.rng_stream(seed, seq, 32, salt_kind=1)
        assert list(raw) == want
