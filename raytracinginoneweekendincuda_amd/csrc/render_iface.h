// render_iface.h -- launch interface between device_scene.cpp (host orchestration) and render.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "flat_scene.h"
#include "rng.h"

namespace rtow {

struct SeedArgs {
    uint32_t *state;             // 6 planes of n_pixels words: d, v0..v4
    const uint32_t *jump_table;  // kJumpTableWords
    Xorwow base;                 // salted seed state (sequence 0)
    uint32_t n_pixels;
    int32_t width, stripe_rows, rank, world_size;
};

struct RenderArgs {
    double *pixels;              // rows_owned x width x 3
    uint32_t *state;
    unsigned long long *ray_counter;
    uint32_t n_pixels;
    int32_t width, height, rows_owned;
    int32_t spp, max_depth;
    int32_t stripe_rows, rank, world_size;
};

hipError_t launch_seed_strict(const SeedArgs &a, hipStream_t stream);
hipError_t launch_seed_fast(const SeedArgs &a, hipStream_t stream);
hipError_t launch_render_strict(const DeviceScene &sc, const RenderArgs &a, hipStream_t stream);
hipError_t launch_render_fast(const DeviceScene &sc, const RenderArgs &a, hipStream_t stream);
hipError_t kernel_attributes_strict(int *vgprs, int *lds_bytes);
hipError_t kernel_attributes_fast(int *vgprs, int *lds_bytes);

} // namespace rtow
