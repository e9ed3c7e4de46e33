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
    uint32_t *tile_cost;        // probe launches: rays traced per 8x8 tile (one counter per tile of this rank's rows)
    const uint32_t *tile_order; // render launches: queue position -> tile (nullptr = tiles in row-major order)
    int32_t probe;              // 1 = cost probe: trace `spp` samples per pixel, write nothing but tile_cost
    uint32_t n_pixels;
    int32_t width, stripe_rows, rank, world_size;
};

struct RenderArgs {
    double *pixels;              // rows_owned x width x 3
    double *accum;               // progressive rendering: running (unnormalised) colour sums, or nullptr
    int32_t spp_before;          // samples already in accum
    uint32_t *state;
    unsigned long long *ray_counter;
    uint32_t *cursor;            // pixel-queue cursor, zeroed before every launch
    uint32_t *tile_cost;        // probe launches: rays traced per 8x8 tile (one counter per tile of this rank's rows)
    const uint32_t *tile_order; // render launches: queue position -> tile (nullptr = tiles in row-major order)
    int32_t probe;              // 1 = cost probe: trace `spp` samples per pixel, write nothing but tile_cost / pix_cost
    // Sphere-list worlds, two classes of pixels (device_scene.cpp rt_render_launch): the probe books every pixel's rays in
    // pix_cost; classify_pixels marks the heavy ones in pix_class and lists them; the frame is then two launches -- the
    // listed pixels (pixel_list, a few lanes-per-ray waves, started first) and all the others (pix_class != 0 is skipped).
    uint32_t *pix_cost;               // probe launches: rays traced by each owned pixel
    uint32_t *dbg_times;              // RT_STAMP builds: per pixel, low words of the wall clock at its start and at its end
    const uint32_t *pixel_list;       // render launch over a list of owned pixels (compact indices) instead of the tile queue
    const uint32_t *pixel_list_count; // device word holding the length of pixel_list
    const uint8_t *pix_class;         // tile-queue launches: pixels whose class is non-zero belong to another launch
    int32_t wave_priority;            // s_setprio for this launch's waves (0..3)
    // The same two classes inside ONE launch (kernels whose workgroup fills a CU: a second launch could not be resident
    // beside it): the first `heavy_waves` waves of every workgroup serve heavy_list (heavy_ppw pixels at a time, through
    // heavy_cursor) and join the tile queue when the list is done; the tile queue skips pix_class != 0 as above.
    const uint32_t *heavy_list;
    const uint32_t *heavy_count;
    // ... the very longest chains among them (classify_pixels: probed cost >= super_threshold) on a list of their own, which the
    // serving waves take from first, ONE pixel per wave and nothing beside it until it is done: the frame cannot end before that
    // pixel does, and alone in its wave its rays take two thirds of the time they take with five neighbours
    const uint32_t *super_list;
    const uint32_t *super_count;
    uint32_t *super_cursor;
    int32_t adaptive_ppw;             // fewer pixels per serving wave where the lists are short (render_kernel: fit)
    int32_t super_ppw;                // pixels of that list per serving wave (1 for BVH walks; sphere lists: 4, the grouped scan's 16 lanes per ray)
    uint32_t *heavy_cursor;
    int32_t heavy_waves, heavy_ppw, heavy_priority;
    uint32_t n_pixels;
    int32_t width, height, rows_owned;
    int32_t spp, max_depth;
    int32_t stripe_rows, rank, world_size;
    int32_t node_burst;     // composite BVH worlds: node visits between two leaf phases (set by the launcher)
    int32_t park_ratio;     // composite BVH worlds: the leaf phase starts once parked lanes outnumber moving ones by this factor
    int32_t leaf_batch;     // composite BVH worlds: a kind of leaf is tested once this many lanes of the wave wait for it
    int32_t rounds;         // kind-batched kernels: node / leaf rounds per look at the shading queue
    int32_t object_batch;   // the same for instances / groups (their cooperative scan serves one ray at a time: a small batch is fine)
    int32_t lds_nodes;      // set by the launcher: BVH nodes are staged in LDS
    int32_t small_world;    // BVH worlds without media are scanned, not walked, up to this scan cost (and 16 leaves)
    int32_t list_waves;     // instanced-list kernel: 4 or 5 waves per SIMD, 0 = by the frame's pixel generations (render.hip list_instances_waves)
    int32_t heavy_scan;     // sphere BVH worlds: the waves that serve the heavy pixels scan all leaves instead of walking
    int32_t accelerate_lists;  // list worlds of primitives: walk the library's tree instead of scanning the list
    int32_t exact_scan;     // sphere-list worlds: no conservative filter in front of the reference's sphere test
    int32_t filter_fp64;    // sphere-list worlds: the fp64 form of that filter, one sphere at a time (default: packed fp32, two at a time)
    int32_t reference_tree; // primitive BVH worlds: walk the reference's own tree in its own order (default: the library's SAH tree)
    int32_t always_walk;    // BVH worlds: walk the tree even where a scan of all leaves would be used (small scenes)
    int32_t force_general;  // tests: use the general kernel even where a specialised one applies
    int32_t coop_threshold; // sphere-list kernel: below this many live lanes a wave scans cooperatively
    int32_t coop_single;    // experiments: cooperative scan one ray at a time (the older scheme) instead of in groups
    int32_t num_cus;
    int32_t lds_spheres;    // set by the launcher: sphere planes staged in LDS for the cooperative scan
    int32_t overdue_priority;
    int32_t boost_rounds;   // overdue-only cooperative passes inserted after each pixel-parallel pass
    int32_t grid_blocks;        // tuning: hard cap on the persistent grid (0 = none)
    int32_t max_blocks_per_cu;  // tuning: cap on resident workgroups per CU (0 = whatever fits)
    int32_t pixels_per_wave;    // sphere-list kernel: at most this many lanes of a wave hold a pixel (64 = all of them)
    int32_t shade_batch;    // BVH kernels: shade once this many lanes have finished their walk
    uint32_t ray_budget;    // sphere-list kernel: a pixel past this many rays is finished cooperatively
};

struct KernelInfo {
    int vgprs, lds_bytes, kind;  // kind = WORLD * 4 + COMPOSITE * 2 + RICH
};

hipError_t launch_seed_strict(const SeedArgs &a, hipStream_t stream);
hipError_t launch_seed_fast(const SeedArgs &a, hipStream_t stream);
hipError_t launch_render_strict(const DeviceScene &sc, const RenderArgs &a, hipStream_t stream);
hipError_t launch_render_fast(const DeviceScene &sc, const RenderArgs &a, hipStream_t stream);
hipError_t kernel_info_strict(const DeviceScene &sc, const RenderArgs &a, KernelInfo *info);
hipError_t kernel_info_fast(const DeviceScene &sc, const RenderArgs &a, KernelInfo *info);

// class 1 + an entry in `list` (its length in *count, which the caller has zeroed) for every pixel whose probed cost is at
// least `threshold` rays, class 0 for the others; with super_list: the pixels of at least super_threshold rays go there instead
// (length count[1])
hipError_t launch_classify_pixels(const uint32_t *pix_cost, uint32_t n_pixels, uint32_t threshold, uint8_t *pix_class, uint32_t *list,
                                  uint32_t *count, hipStream_t stream, uint32_t *super_list = nullptr, uint32_t super_threshold = 0,
                                  uint32_t width = 0, uint32_t near_percent = 0, uint32_t near_neighbours = 0);  // a pixel of near_percent % of the threshold with that many of its 8 neighbours over it is listed too

// tile_order[k] = the tile with the k-th highest cost (counting sort over 256 cost classes; one workgroup)
// (flat_x8 / 8 = ratio of the heaviest tile to the mean below which the row-major order is kept)
hipError_t launch_tile_order(const uint32_t *tile_cost, uint32_t *tile_order, uint32_t n_tiles, uint32_t flat_x8, hipStream_t stream);

} // namespace rtow
