// rng.h -- per-pixel RNG shared by host scene generation and the HIP kernels.
//
// cuRAND XORWOW device API semantics (the reference's curandState / curand_init / curand_uniform,
// call sites R/kernel.cu:105,118,140-141, R/Material.h:19-21, R/Camera.h:15,80, R/Dielectric.h:41,
// R/ConstantMedium.h:79, R/Perlin.h:91-93,109).  cuRAND itself is third-party and absent from the
// reference tree; this is a from-scratch statement of the published algorithm:
//   state  : five 32-bit xorshift words + a Weyl counter d
//   step   : t = v0 ^ (v0 >> 2); shift words down; v4 = (v4 ^ (v4 << 4)) ^ (t ^ (t << 1)); d += 362437
//   output : v4 + d
//   init   : seed salted into (v, d); sequence n = skip n * 2^67 steps (a GF(2)-linear map on v; d unchanged)
//   uniform: float(x) * 2^-32 + 2^-33 in fp32  ->  (0, 1]
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtow {

struct Xorwow {
    uint32_t d, v0, v1, v2, v3, v4;
};

RT_HD uint32_t xorwow_next(Xorwow &s)
{
    uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1;
    s.v1 = s.v2;
    s.v2 = s.v3;
    s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}

// curand_uniform: both constants are exact powers of two, so the fp32 result is a single rounding of
// x * 2^-32 + 2^-33 whether or not the multiply-add is fused.
RT_HD float xorwow_uniform(Xorwow &s)
{
    return (float)xorwow_next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

// Seed salting of curand_init (cuRAND device API constants); salt = {xor0, xor1, mul0, mul1}.
struct XorwowSalt {
    uint32_t x0, x1, m0, m1;
};
constexpr XorwowSalt kSaltCurandDevice = {0xaad26b49u, 0xf7dcefddu, 1099087573u, 2591861531u};
constexpr XorwowSalt kSaltRocrand = {0x2c7f967fu, 0xa03697cbu, 1228688033u, 2073658381u}; // tests only

RT_HD Xorwow xorwow_seed(uint64_t seed, const XorwowSalt salt)
{
    uint32_t s0 = (uint32_t)seed ^ salt.x0;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ salt.x1;
    uint32_t t0 = salt.m0 * s0;
    uint32_t t1 = salt.m1 * s1;
    Xorwow s;
    s.d = 6615241u + t1 + t0;
    s.v0 = 123456789u + t0;
    s.v1 = 362436069u ^ t0;
    s.v2 = 521288629u + t1;
    s.v3 = 88675123u ^ t1;
    s.v4 = 5783321u + t0;
    return s;
}

// Sequence jump tables: radix-16 digits of the sequence number.  Entry [k][g-1] (g = 1..15) is the
// 160x160 GF(2) matrix T^(2^67 * g * 16^k), stored as 160 rows (one per input bit) of 5 words.
constexpr int kJumpDigits = 16;  // 16 hex digits cover a 64-bit sequence number
constexpr int kJumpRowWords = 5;
constexpr int kJumpMatrixWords = 160 * kJumpRowWords;
constexpr size_t kJumpTableWords = (size_t)kJumpDigits * 15 * kJumpMatrixWords;

RT_HD void xorwow_apply(const uint32_t *m /* 160 x 5 */, Xorwow &s)
{
    const uint32_t in[5] = {s.v0, s.v1, s.v2, s.v3, s.v4};
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    for (int w = 0; w < 5; w++) {
        uint32_t bits = in[w];
        const uint32_t *rows = m + (size_t)w * 32 * kJumpRowWords;
        while (bits) {
            int b = __builtin_ctz(bits);
            bits &= bits - 1;
            const uint32_t *r = rows + b * kJumpRowWords;
            a0 ^= r[0]; a1 ^= r[1]; a2 ^= r[2]; a3 ^= r[3]; a4 ^= r[4];
        }
    }
    s.v0 = a0; s.v1 = a1; s.v2 = a2; s.v3 = a3; s.v4 = a4;
}

// curand_init(seed, sequence, 0): jump the salted seed state forward by sequence * 2^67 steps.
RT_HD void xorwow_skip_sequences(const uint32_t *table, uint64_t sequence, Xorwow &s)
{
    for (int k = 0; sequence != 0; k++, sequence >>= 4) {
        uint32_t g = (uint32_t)sequence & 15u;
        if (g) xorwow_apply(table + ((size_t)k * 15 + (g - 1)) * kJumpMatrixWords, s);
    }
}

} // namespace rtow
