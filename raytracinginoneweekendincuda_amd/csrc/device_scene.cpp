// device_scene.cpp -- HIP side of the render API (include/rtow.h): table upload, the film (framebuffer +
// per-pixel RNG state, the reference's frameBuffer / randState, R/kernel.cu:606-613), launches and
// timing.  Replaces the host driver section R/kernel.cu:675-691.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rtow.h"
#include "render_iface.h"
#include "scene_host.h"

namespace rtow {

static int hip_fail(hipError_t e, const char *what)
{
    return fail(RT_ERR_HIP, std::string("HIP error = ") + std::to_string((unsigned)e) + " (" + hipGetErrorString(e) +
                                ") at '" + what + "'");
}
#define HIP_TRY(expr)                                     \
    do {                                                  \
        hipError_t e_ = (expr);                           \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

struct DeviceTables {
    int device = -1;
    uint64_t generation = 0;  // SceneImpl::generation these tables were made from
    std::vector<void *> allocations;
    DeviceScene scene{};
};

void release_device_tables(DeviceTables *t)
{
    if (!t) return;
    int prev = 0;
    hipGetDevice(&prev);
    hipSetDevice(t->device);
    for (void *p : t->allocations) hipFree(p);
    hipSetDevice(prev);
    delete t;
}

template <class T>
static int upload(DeviceTables &dt, const std::vector<T> &host, const T *&dev)
{
    dev = nullptr;
    // keep every table pointer valid (never null) so that speculative loads stay in bounds
    size_t bytes = (host.empty() ? 1 : host.size()) * sizeof(T);
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    dt.allocations.push_back(p);
    if (!host.empty())
        HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    else
        HIP_TRY(hipMemset(p, 0, bytes));
    dev = static_cast<const T *>(p);
    return RT_OK;
}

// jump table: one copy per device for the life of the process
static std::mutex g_jump_mutex;
static std::map<int, uint32_t *> g_jump_tables;
static int device_jump_table(int device, const uint32_t **out)
{
    std::lock_guard<std::mutex> lock(g_jump_mutex);
    auto it = g_jump_tables.find(device);
    if (it == g_jump_tables.end()) {
        uint32_t *p = nullptr;
        HIP_TRY(hipMalloc((void **)&p, kJumpTableWords * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(p, host_jump_table(), kJumpTableWords * sizeof(uint32_t), hipMemcpyHostToDevice));
        it = g_jump_tables.emplace(device, p).first;
    }
    *out = it->second;
    return RT_OK;
}

static int select_device(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(RT_ERR_NO_DEVICE, "no HIP device available: the gfx950 render path cannot run (there is no CPU fallback)");
    if (device < 0 || device >= count) return fail(RT_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    return RT_OK;
}

constexpr size_t kCounterWords = 128;

// Scheduling constants are fixed in the shipping library: it reads nothing from the environment on the launch path.
// A/B builds (make EXTRA=-DRT_TUNING=1 LIBNAME=..., loaded through RTOW_LIB_PATH by the Python binding) compile the
// overrides in; every override is clamped to the range the kernels accept (a zero pixels-per-wave or round count would
// leave the persistent waves spinning).  None of them changes an image.
#ifndef RT_TUNING
#define RT_TUNING 0
#endif
#if RT_TUNING
static int tune(const char *name, int value, int lo, int hi)
{
    if (const char *e = std::getenv(name)) {
        const int v = std::atoi(e);
        value = v < lo ? lo : (v > hi ? hi : v);
    }
    return value;
}
static bool tune_set(const char *name) { return std::getenv(name) != nullptr; }
#else
static inline int tune(const char *, int value, int, int) { return value; }
static inline bool tune_set(const char *) { return false; }
#endif

struct FilmImpl {
    int device = 0;
    int width = 0, height = 0, stripe_rows = 8, rank = 0, world_size = 1;
    int rows_owned = 0;
    uint32_t n_pixels = 0;
    double *pixels = nullptr;      // where the kernel writes (own_pixels or a bound external buffer)
    double *own_pixels = nullptr;
    double *accum = nullptr;       // progressive rendering: unnormalised colour sums (allocated on first use)
    int accum_spp = 0;
    uint32_t *state = nullptr;
    unsigned long long *ray_counter = nullptr;  // [0] rays, [1] low word = pixel-queue cursor, [2..5] stamps, [7] and [16..63] phase sums
    int num_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t last_stream = nullptr;
    uint32_t *tile_cost = nullptr, *tile_order = nullptr;  // per 8x8 tile of this rank's rows: probed rays, and the tiles ranked by them
    // sphere-list worlds, heavy / light pixels (allocated on first use): probed rays per pixel, the heavy pixels' list and
    // count, every pixel's class; the heavy launch runs on its own stream beside the light one
    uint32_t *pix_cost = nullptr, *heavy_list = nullptr, *heavy_count = nullptr, *super_list = nullptr;  // heavy_count[1]: length of super_list
    uint32_t *dbg_times = nullptr;  // RT_STAMP diagnostic builds (RTOW_PRINT_TAIL)
    uint8_t *pix_class = nullptr;
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_aux[2] = {nullptr, nullptr};
    uint32_t n_tiles = 0;
    unsigned long long *host_counters = nullptr;  // pinned mirror of ray_counter, filled by an async copy behind the render
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // begin, after seed, after render, after the counter copy
    bool seeded = false;
    bool in_flight = false;
    SceneImpl *scene_in_flight = nullptr;  // the scene whose tables the launch in flight reads
    uint64_t last_samples = 0;
    int last_variant = 0;
    KernelInfo last_kernel{};
    int last_pixels_per_wave = 64;
};

// a launch of `s` into `f` is (about to be) in flight / is over
static void mark_in_flight(SceneImpl &s, FilmImpl &f)
{
    f.in_flight = true;
    f.scene_in_flight = &s;
    s.launches_in_flight++;
    s.films_in_flight.push_back(&f);
}
static void mark_done(FilmImpl &f)
{
    f.in_flight = false;
    if (SceneImpl *s = f.scene_in_flight) {
        s->launches_in_flight--;
        for (size_t k = 0; k < s->films_in_flight.size(); k++)
            if (s->films_in_flight[k] == &f) {
                s->films_in_flight.erase(s->films_in_flight.begin() + (long)k);
                break;
            }
        f.scene_in_flight = nullptr;
    }
}

void wait_for_films_in_flight(SceneImpl &s)
{
    int prev = 0;
    hipGetDevice(&prev);
    for (void *p : s.films_in_flight) {
        FilmImpl *f = static_cast<FilmImpl *>(p);
        hipSetDevice(f->device);
        // the whole stream, not only the last event: a launch that failed half-way has recorded no event
        if (f->last_stream) hipStreamSynchronize(f->last_stream);
        else if (f->ev[3]) hipEventSynchronize(f->ev[3]);
        f->scene_in_flight = nullptr;  // the film stays "in flight" until its rt_render_finish, which then only reports
    }
    hipSetDevice(prev);
    s.films_in_flight.clear();
    s.launches_in_flight = 0;
}

} // namespace rtow

using namespace rtow;

static inline SceneImpl *S(rt_scene *s) { return reinterpret_cast<SceneImpl *>(s); }
static inline FilmImpl *F(rt_film *f) { return reinterpret_cast<FilmImpl *>(f); }

extern "C" {

int rt_scene_upload(rt_scene *scene, int device)
{
    if (!scene) return fail(RT_ERR_INVALID, "rt_scene_upload: null scene");
    SceneImpl &s = *S(scene);
    if (!s.committed) return fail(RT_ERR_STATE, "rt_scene_upload: scene not committed (rt_scene_commit)");
    if (int rc = select_device(device)) return rc;
    if ((int)s.device.size() <= device) s.device.resize(device + 1, nullptr);
    if (s.device[device]) {
        if (s.device[device]->generation == s.generation) return RT_OK;
        // The scene was re-committed or its camera changed since this copy was made.  Nothing can still be reading
        // it: commit / set_camera refuse while a launch is in flight.
        release_device_tables(s.device[device]);
        s.device[device] = nullptr;
    }
    DeviceTables *dt = new DeviceTables;
    dt->device = device;
    dt->generation = s.generation;
    const FlatScene &f = s.flat;
    DeviceScene &d = dt->scene;
    int rc = RT_OK;
    auto up = [&](auto &host, auto &dev) {
        if (rc == RT_OK) rc = upload(*dt, host, dev);
    };
    up(f.spheres, d.spheres);
    up(f.sphere_scan, d.sphere_scan);
    up(f.sphere_scan32, d.sphere_scan32);
    up(f.sphere_aux, d.sphere_aux);
    up(f.mspheres, d.mspheres);
    up(f.ms_planes, d.ms_planes);
    up(f.msphere_aux, d.msphere_aux);
    up(f.quads, d.quads);
    up(f.quad_aa, d.quad_aa);
    up(f.boxes, d.boxes);
    up(f.quad_mat, d.quad_mat);
    up(f.objects, d.objects);
    up(f.items, d.items);
    up(f.xforms, d.xforms);
    up(f.media, d.media);
    up(f.group_boxes, d.group_boxes);
    up(f.tree_nodes, d.tree_nodes);
    up(f.tree_items, d.tree_items);
    up(f.tree_bvh, d.tree_bvh);
    up(f.nodes, d.nodes);
    up(f.fast_nodes, d.fast_nodes);
    up(f.fast_order, d.fast_order);
    {
        // The library tree's rows as the kernels read them: boxes in fp32, outwards (flat_scene.h FastNodeF).  The widening covers
        // what a conservative fp32 slab test can lose: the rounding of the box, of the ray's origin and of the products, each at
        // most 2^-23 of the magnitudes involved -- the scene's reach (every leaf box, the media's included: rays start on
        // surfaces, inside media or at the camera).
        double reach = 1.0;
        for (const Box &b : f.leaf_boxes)
            for (int a = 0; a < 3; a++) reach = std::fmax(reach, std::fmax(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
        for (int a = 0; a < 3; a++) reach = std::fmax(reach, std::fabs(s.camera.origin[a]) + std::fabs(s.camera.lens_radius) * 2.0);
        const double widen = reach * 1.9073486328125e-06;  // 2^-19
        auto down = [](double v) { float x = (float)v; return (double)x > v ? std::nextafterf(x, -INFINITY) : x; };
        auto up_f = [](double v) { float x = (float)v; return (double)x < v ? std::nextafterf(x, INFINITY) : x; };
        std::vector<FastNodeF> rows(f.fast_nodes.size());
        for (size_t k = 0; k < rows.size(); k++) {
            const FastNodeRec &n = f.fast_nodes[k];
            FastNodeF &r = rows[k];
            const double lo[3] = {n.xlo, n.ylo, n.zlo}, hi[3] = {n.xhi, n.yhi, n.zhi};
            for (int a = 0; a < 3; a++) {
                r.box[2 * a] = down(lo[a] - widen);
                r.box[2 * a + 1] = up_f(hi[a] + widen);
            }
            r.a = n.a;
            r.b = n.b;
            std::memcpy(r.link, n.link, sizeof r.link);
            r.pad = 0;
        }
        up(rows, d.fast_rows);
        std::vector<SegMedium> media = f.seg_media;
        for (SegMedium &m : media)
            for (int a = 0; a < 3; a++) {
                m.fbox[2 * a] = down(m.lo[a] - widen);
                m.fbox[2 * a + 1] = up_f(m.hi[a] + widen);
            }
        up(media, d.seg_media);
    }
    up(f.seg_cand, d.seg_cand);
    up(f.world_items, d.world_items);
    up(f.materials, d.materials);
    up(f.textures, d.textures);
    up(f.images, d.images);
    up(f.image_bytes, d.image_bytes);
    up(f.perlin, d.perlin);
    std::vector<CameraRec> cam_host(1, s.camera);
    up(cam_host, d.camera);
    if (rc != RT_OK) {
        release_device_tables(dt);
        return rc;
    }
    d.world_kind = f.world_kind;
    d.n_world_items = (uint32_t)f.world_items.size();
    d.n_nodes = (uint32_t)f.nodes.size();
    d.n_world_nodes = f.n_world_nodes;
    d.scan_cost = f.scan_cost;
    d.n_spheres = (uint32_t)f.spheres.size();
    d.scan_reach = f.scan_reach;
    d.scan_reach32 = f.scan_reach32;
    d.n_mspheres = (uint32_t)f.mspheres.size();
    d.n_quads = (uint32_t)f.quads.size();
    d.n_objects = (uint32_t)f.objects.size();
    d.n_boxes = (uint32_t)f.boxes.size();
    d.n_xforms = (uint32_t)f.xforms.size();
    if (!(f.flags & SCENE_WORLD_MSPHERES)) d.ms_planes = nullptr;
    d.ms_padded = f.ms_padded;
    d.n_fast_nodes = (uint32_t)f.fast_nodes.size();
    if (f.fast_nodes.empty()) d.fast_nodes = nullptr;
    d.n_seg_media = (uint32_t)f.seg_media.size();
    d.n_seg_cand = (uint32_t)f.seg_cand.size();
    d.lds_fast_order = d.lds_seg_media = d.lds_seg_cand = kNone;
    d.n_media = (uint32_t)f.media.size();
    d.n_materials = (uint32_t)f.materials.size();
    d.n_perlin = (uint32_t)f.perlin.size();
    d.n_group_boxes = (uint32_t)f.group_boxes.size();
    d.lds_quad_aa = d.lds_boxes = d.lds_objects = d.lds_xforms = d.lds_media = d.lds_materials = d.lds_perlin = d.lds_spheres_tab = d.lds_group_boxes = kNone;
    d.lds_mspheres = d.lds_msphere_aux = d.lds_sphere_aux = kNone;
    d.flags = f.flags;
    s.device[device] = dt;
    return RT_OK;
}

rt_film *rt_film_create(int device, int width, int height, int stripe_rows, int rank, int world_size)
{
    if (width <= 0 || height <= 0 || stripe_rows <= 0 || world_size <= 0 || rank < 0 || rank >= world_size) {
        set_error("rt_film_create: bad geometry");
        return nullptr;
    }
    if ((int64_t)width * height >= (int64_t)1 << 31) {
        set_error("rt_film_create: frame too large (pixel index must fit in int like the reference's pixelIndex)");
        return nullptr;
    }
    if (select_device(device) != RT_OK) return nullptr;
    FilmImpl *f = new FilmImpl;
    f->device = device;
    f->width = width;
    f->height = height;
    f->stripe_rows = stripe_rows;
    f->rank = rank;
    f->world_size = world_size;
    f->rows_owned = rt_stripe_rows(height, stripe_rows, rank, world_size, nullptr, 0);
    f->n_pixels = (uint32_t)f->rows_owned * (uint32_t)width;
    size_t np = f->n_pixels ? f->n_pixels : 1;
    hipError_t e = hipMalloc((void **)&f->own_pixels, np * 3 * sizeof(double));
    if (e == hipSuccess) e = hipMemset(f->own_pixels, 0, np * 3 * sizeof(double));
    f->pixels = f->own_pixels;
    if (e == hipSuccess) e = hipMalloc((void **)&f->state, np * 6 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&f->ray_counter, kCounterWords * sizeof(unsigned long long));
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        if (e == hipSuccess) f->num_cus = prop.multiProcessorCount;
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&f->own_stream, hipStreamNonBlocking);
    f->n_tiles = (((uint32_t)width + 7u) >> 3) * (((uint32_t)f->rows_owned + 7u) >> 3);
    if (e == hipSuccess) e = hipMalloc((void **)&f->tile_cost, (f->n_tiles ? f->n_tiles : 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&f->tile_order, (f->n_tiles ? f->n_tiles : 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&f->host_counters, kCounterWords * sizeof(unsigned long long), hipHostMallocDefault);
    for (int k = 0; k < 4 && e == hipSuccess; k++) e = hipEventCreate(&f->ev[k]);
    if (e != hipSuccess) {
        hip_fail(e, "rt_film_create allocation");
        rt_film_destroy(reinterpret_cast<rt_film *>(f));
        return nullptr;
    }
    return reinterpret_cast<rt_film *>(f);
}

void rt_film_destroy(rt_film *film)
{
    if (!film) return;
    FilmImpl *f = F(film);
    hipSetDevice(f->device);
    if (f->in_flight) {  // destroyed without rt_render_finish: wait for the kernel, give the scene its count back
        if (f->last_stream) hipStreamSynchronize(f->last_stream);
        mark_done(*f);
    }
    if (f->own_pixels) hipFree(f->own_pixels);
    if (f->accum) hipFree(f->accum);
    if (f->state) hipFree(f->state);
    if (f->ray_counter) hipFree(f->ray_counter);
    if (f->tile_cost) hipFree(f->tile_cost);
    if (f->tile_order) hipFree(f->tile_order);
    if (f->pix_cost) hipFree(f->pix_cost);
    if (f->dbg_times) hipFree(f->dbg_times);
    if (f->heavy_list) hipFree(f->heavy_list);
    if (f->heavy_count) hipFree(f->heavy_count);
    if (f->super_list) hipFree(f->super_list);
    if (f->pix_class) hipFree(f->pix_class);
    for (int k = 0; k < 2; k++)
        if (f->ev_aux[k]) hipEventDestroy(f->ev_aux[k]);
    if (f->aux_stream) hipStreamDestroy(f->aux_stream);
    if (f->host_counters) hipHostFree(f->host_counters);
    for (int k = 0; k < 4; k++)
        if (f->ev[k]) hipEventDestroy(f->ev[k]);
    if (f->own_stream) hipStreamDestroy(f->own_stream);
    delete f;
}

void *rt_film_device_pixels(rt_film *film) { return film ? F(film)->pixels : nullptr; }
size_t rt_film_pixel_bytes(rt_film *film) { return film ? (size_t)F(film)->n_pixels * 3 * sizeof(double) : 0; }
int rt_film_bind_pixels(rt_film *film, void *device_pixels)
{
    if (!film) return fail(RT_ERR_INVALID, "rt_film_bind_pixels: null film");
    if (F(film)->in_flight) return fail(RT_ERR_STATE, "rt_film_bind_pixels: a render is in flight");
    F(film)->pixels = device_pixels ? static_cast<double *>(device_pixels) : F(film)->own_pixels;
    return RT_OK;
}

// Everything rt_render_launch puts on the stream: seeding, the rehearsal and its bookkeeping, the render kernel, the
// counter copy.  Called with the film already marked in flight (a failure half-way leaves kernels running).
static int enqueue_frame(SceneImpl &s, FilmImpl &f, const rt_render_params *p, hipStream_t stream)
{
    const bool keep = (p->flags & RT_FLAG_KEEP_RNG_STATE) && f.seeded;

    HIP_TRY(hipMemsetAsync(f.ray_counter, 0, kCounterWords * sizeof(unsigned long long), stream));
    HIP_TRY(hipMemsetAsync(f.ray_counter + 2, 0xFF, sizeof(unsigned long long), stream));  // stamp slots (diagnostic builds): min
    HIP_TRY(hipMemsetAsync(f.ray_counter + 4, 0xFF, 2 * sizeof(unsigned long long), stream));
    HIP_TRY(hipEventRecord(f.ev[0], stream));
    if (!keep) {
        SeedArgs sa{};
        sa.state = f.state;
        if (int rc = device_jump_table(f.device, &sa.jump_table)) return rc;
        sa.base = xorwow_seed(p->seed, kSaltCurandDevice);
        sa.n_pixels = f.n_pixels;
        sa.width = f.width;
        sa.stripe_rows = f.stripe_rows;
        sa.rank = f.rank;
        sa.world_size = f.world_size;
        HIP_TRY(p->variant ? launch_seed_fast(sa, stream) : launch_seed_strict(sa, stream));
        f.seeded = true;
    }
    HIP_TRY(hipEventRecord(f.ev[1], stream));
    RenderArgs ra{};
    ra.pixels = f.pixels;
    if (p->flags & RT_FLAG_ACCUMULATE) {
        // progressive frame: this launch's samples are added to the film's running sums (needs the saved RNG streams)
        if (!f.accum) {
            HIP_TRY(hipMalloc((void **)&f.accum, (size_t)(f.n_pixels ? f.n_pixels : 1) * 3 * sizeof(double)));
            f.accum_spp = 0;
        }
        if (!keep) f.accum_spp = 0;  // re-seeded: start a new frame
        ra.accum = f.accum;
        ra.spp_before = f.accum_spp;
        if (p->samples_per_pixel > 0) f.accum_spp += p->samples_per_pixel;
    }
    ra.state = f.state;
    ra.ray_counter = f.ray_counter;
    ra.cursor = reinterpret_cast<uint32_t *>(f.ray_counter + 1);
    ra.coop_threshold = p->coop_threshold > 0 ? p->coop_threshold : 24;
    ra.coop_single = (p->flags & RT_FLAG_COOP_SINGLE) ? 1 : 0;
    ra.num_cus = f.num_cus;
    ra.shade_batch = p->shade_batch > 0 ? p->shade_batch : 16;
    ra.max_blocks_per_cu = p->max_blocks_per_cu;
    ra.pixels_per_wave = 64;  // settled below, once the kernel is known
    ra.boost_rounds = tune("RTOW_BOOST", 8, 0, 1024);
    ra.grid_blocks = tune("RTOW_GRID_BLOCKS", 0, 0, 1 << 20);
    if (ra.grid_blocks > 0) ra.max_blocks_per_cu = 8;
    // A deep world BVH over composite leaves (scene 9: 400 boxes, two media, an instanced cluster): a leaf phase costs
    // tens of node steps there, so it pays to wait until most walkers have parked.  A shallow one (Cornell box: 8
    // leaves) gains nothing from waiting.
    {
        const uint32_t world_nodes = s.flat.n_world_nodes;
        ra.node_burst = tune("RTOW_BURST", world_nodes > 64 ? 24 : 8, 1, 4096);
        ra.park_ratio = tune("RTOW_PARK", world_nodes > 64 ? 4 : 1, 1, 64);
        ra.leaf_batch = tune("RTOW_LEAF_BATCH", 12, 1, 64);
        ra.object_batch = tune("RTOW_OBJECT_BATCH", 4, 1, 64);
        ra.rounds = tune("RTOW_ROUNDS", 4, 1, 64);
    }
    ra.overdue_priority = (p->flags & RT_FLAG_OVERDUE_PRIORITY) ? 1 : 0;
    {
        // Off by default: on the Book-1 scenes a cooperative ray costs ~10x a pixel-parallel one, and every budget
        // tried (3..16 rays/sample, 2..32 boost rounds) lost more in throughput than it won back in frame tail.
        double per_sample = p->overdue_rays_per_sample > 0 ? (double)p->overdue_rays_per_sample : 1.0e9;
        double budget = per_sample * (double)p->samples_per_pixel;
        ra.ray_budget = budget >= 4.0e9 ? 0xFFFFFFFFu : (uint32_t)budget;
        if (p->overdue_rays_per_sample < 0) ra.ray_budget = 0xFFFFFFFFu;  // negative: never
    }
    ra.n_pixels = f.n_pixels;
    ra.width = f.width;
    ra.height = f.height;
    ra.rows_owned = f.rows_owned;
    ra.spp = p->samples_per_pixel;
    ra.max_depth = p->max_depth;
    ra.stripe_rows = f.stripe_rows;
    ra.rank = f.rank;
    ra.world_size = f.world_size;
    ra.force_general = (p->flags & RT_FLAG_FORCE_GENERAL) ? 1 : 0;
    ra.always_walk = (p->flags & RT_FLAG_ALWAYS_WALK) ? 1 : 0;
    ra.reference_tree = (p->flags & RT_FLAG_REFERENCE_TREE) ? 1 : 0;
    ra.exact_scan = tune("RTOW_EXACT_SCAN", (p->flags & RT_FLAG_EXACT_SCAN) ? 1 : 0, 0, 1);
    ra.accelerate_lists = (p->flags & RT_FLAG_ACCELERATE_LISTS) ? 1 : 0;
    ra.filter_fp64 = tune("RTOW_FILTER_FP64", (p->flags & RT_FLAG_FILTER_FP64) ? 1 : 0, 0, 1);
    ra.heavy_scan = tune("RTOW_HEAVY_SCAN", 0, 0, 1);
    ra.list_waves = tune("RTOW_LIST_WAVES", 0, 0, 5);
    ra.small_world = tune("RTOW_SMALL_WORLD", 64, 0, 1 << 20);  // scan budget in half sphere tests, see FlatScene::scan_cost
    const DeviceScene &ds = s.device[f.device]->scene;
    HIP_TRY(p->variant ? kernel_info_fast(ds, ra, &f.last_kernel) : kernel_info_strict(ds, ra, &f.last_kernel));
    const int kind = f.last_kernel.kind & 63;
    // BVH sphere worlds: thin waves may scan all leaves together instead of walking (scan_grouped_ms), but the planes
    // come from L2 and a chip full of thin waves scanning is bound by L2 bandwidth: measured slower than walking at every
    // threshold (C3: 1748 Msamples/s never, 1681 at 17, 1048 at 33).  Off unless asked for.
    if (kind < 8 && p->coop_threshold <= 0) ra.coop_threshold = 0;
    // A pixel's samples are one sequential chain (one RNG stream), so a frame cannot end before its longest pixel does
    // (glass: up to max_depth rays per sample).  One rehearsal of the first samples of every pixel -- the same RNG streams,
    // nothing written but ray counts, cost probe_spp / spp of the frame -- serves two schedulers:
    //  * BVH worlds, heaviest tiles first: the 8x8 tiles are ranked by rays traced and the pixel queue hands them out in
    //    that order (in row-major order C3's queue drained at 38 ms and the last wave left at 99 ms);
    //  * sphere-list and primitive-BVH worlds, heavy and light pixels: the few pixels with long chains (0.4 % of C2's
    //    trace more than 10 rays per sample, up to 41) are listed; two waves of every workgroup serve that list first, a
    //    few pixels at a time -- the lanes share each ray's scan (sphere list: a third of the latency per ray at 1.8x the
    //    work), or simply have the wave to themselves (BVH walk) -- and then join the tile queue, whose pixels skip the
    //    listed ones.  (The first form, a launch of its own for the list on a second stream, is still there for
    //    tuning builds, RTOW_ROLES=0.)  C2 took 367 ms where its throughput alone needs ~310.
    // Every pixel is still rendered exactly once from its own stream: the frame is the same bit for bit
    // (tests: ...tile_ranking..., ...heavy_and_light...; RT_FLAG_ROW_MAJOR_TILES / RT_FLAG_NO_PIXEL_CLASSES turn them off).
    const bool bvh_kernel = kind < 8;
    const bool sphere_list_kernel = kind >= 16 && kind < 32, prim_bvh_kernel = kind == 0;
    const bool list_scan_kernel = kind == 8 || kind == 10;  // list scans without media: leaves can be dealt to lanes (render.hip scan_leaves_grouped)
    //  * list worlds too (r3): a frame is a few "generations" of pixels per lane (C4: 640 k pixels on 262 k lanes), and the last
    //    generation lasts as long as its longest pixel while ever fewer lanes are busy (C4: queue dry at 148 ms, last wave out
    //    at 261).  Heaviest tiles first makes the pixels that start last the cheap ones.
    bool rank_tiles = (bvh_kernel || list_scan_kernel || sphere_list_kernel) && p->samples_per_pixel >= 32 && f.n_tiles >= 1024 &&
                      !(p->flags & RT_FLAG_ROW_MAJOR_TILES);
    rank_tiles = rank_tiles && tune("RTOW_TILE_SORT", 1, 0, 1) != 0;
    // the deep general kernel (one 768-thread workgroup per CU, C5): its ray chains are the longest of all (a ray takes ~140 us
    // in a full wave), which decides the frame whenever a GPU holds few pixels per lane -- a small frame, or one rank's share
    const bool deep_kernel = kind == 7 && f.last_kernel.lds_bytes > 64 * 1024;
    // (r3) Heavy and light pixels for this kernel too -- where the frame is a few generations of pixels on the GPU's lanes, i.e. a
    // rank's stripes of a split frame.  C5 at 200 spp, one rank of N rendered alone: 699 / 712 / 646 / 597 / 709 / 479 ms for
    // N = 2 / 3 / 4 / 6 / 8 / 16 without classes -- no scaling at all, every rank waits for its longest chains -- and
    // 737 / 566 / 447 / 395 / 368 / 302 ms with them (profiles/r03_c5_roles.txt; the 8-way figure 293 with the settings of the
    // second sweep there: ten of twelve waves serving 32 pixels each from 8 rays per sample -- most of the frame, in half-filled waves).  A whole frame (13 generations) is throughput
    // and loses by them (1030 -> 1400 ms and worse), as it did in r2 with other settings; so: up to seven generations (a 2-way
    // split, 672 -> 626 ms), in three bands of settings.
    const double generations = (double)f.n_pixels / ((double)f.num_cus * 12.0 * 64.0);
    const bool deep_roles = deep_kernel && tune("RTOW_ROLES_DEEP", generations <= 7.0 ? 1 : 0, 0, 1) != 0;
    const bool ppw_given = (p->pixels_per_wave > 0 && p->pixels_per_wave < 64) || tune_set("RTOW_PIXELS_PER_WAVE");
    bool split = (sphere_list_kernel || prim_bvh_kernel || deep_roles) && !(p->flags & RT_FLAG_NO_PIXEL_CLASSES) && p->samples_per_pixel >= 64 &&
                 f.n_pixels >= 65536u && !ppw_given;
    split = split && tune("RTOW_PIXEL_CLASSES", 1, 0, 1) != 0;
    // The primitive-BVH kernel on the reference's tree (256-thread workgroups) gained nothing from a second launch (C3
    // 1672 -> 1100-1200: opt-in); the library-tree kernel, whose 768-thread workgroup fills a CU, serves both classes in
    // ONE launch, by wave (RenderArgs::heavy_list): C3 2106 -> 2713 Msamples/s.
    // Sphere-list worlds: both forms work; serving the heavy pixels from two waves of every workgroup lets the launch keep
    // three workgroups per CU resident (a third wave per SIMD: +15 % in the steady state, which a frame whose end is set by
    // its long pixels could not use) -- C2 1479 (two launches, two workgroups per CU) -> 1556 Msamples/s.
    bool roles_in_one_launch = (prim_bvh_kernel && (f.last_kernel.kind & 64) != 0) || deep_roles || sphere_list_kernel;
    if (prim_bvh_kernel && !roles_in_one_launch && !tune_set("RTOW_PIXEL_CLASSES")) split = false;
    roles_in_one_launch = tune("RTOW_ROLES", roles_in_one_launch ? 1 : 0, 0, 1) != 0;

    // pixels_per_wave < 64 gives every ray several lanes: the sphere list deals a ray's spheres to the lanes of a group (its
    // heavy-pixel waves do that by themselves, above), the list-scan kernels a ray's leaves (render.hip scan_leaves_grouped).
    // 0 = the library's choice, and that is 64 for every frame size measured: a pixel's samples are one sequential chain, a
    // launch ends with its longest pixel, and the pass of a wave that holds a few rays takes as long as one that holds 64 --
    // 17.6 us on the Cornell box whether the film owns 640 k pixels or 5 k (1 / 128 of C4: 118 ms for every split from 1 / 8
    // on) -- while the grouped pass is LONGER, not shorter: the eight leaves of that world are three kinds of code, which a
    // group of lanes executes one after the other just as one lane does, plus the exchange (C4 / 8: 117 ms at 64 pixels per
    // wave, 182 at 32, 201 at 16, 324 at 8; profiles/r03_lanes_per_ray.txt).  The parameter stays for worlds of one kind of
    // leaf and for experiments; the frames are bit-identical either way.
    {
        int ppw = 64;
        if (p->pixels_per_wave > 0 && p->pixels_per_wave < 64) ppw = p->pixels_per_wave;
        ppw = tune("RTOW_PIXELS_PER_WAVE", ppw, 1, 64);
        if (list_scan_kernel) {  // the grouped leaf scan deals lanes in powers of two
            int pow2 = 1;
            while (pow2 * 2 <= ppw) pow2 *= 2;
            ppw = pow2;
        }
        ra.pixels_per_wave = ppw;
    }
    f.last_pixels_per_wave = ra.pixels_per_wave;
    if (ra.pixels_per_wave < 64 && list_scan_kernel)  // the instantiation that deals leaves to lanes: report that one
        HIP_TRY(p->variant ? kernel_info_fast(ds, ra, &f.last_kernel) : kernel_info_strict(ds, ra, &f.last_kernel));

    if (tune_set("RTOW_PRINT_TAIL")) {  // diagnostic builds (-DRT_STAMP=1 -DRT_TUNING=1)
        if (!f.dbg_times) HIP_TRY(hipMalloc((void **)&f.dbg_times, (size_t)f.n_pixels * 2 * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(f.dbg_times, 0, (size_t)f.n_pixels * 2 * sizeof(uint32_t), stream));
        ra.dbg_times = f.dbg_times;
    }
    if (rank_tiles || split) {
        // sphere-list frames of 400 samples and more rehearse 8: the heavy pixels are told apart more reliably (C2, three
        // interleaved pairs in one call: 1859-1893 with 4, 1908-1918 with 8; the primitive-BVH kernel is better off with 4)
        int probe_spp = split ? (p->samples_per_pixel >= 400 ? 8 : 4) : p->samples_per_pixel / 100;  // (r3: 8 for the BVH kernel too, with the settings below)
        probe_spp = probe_spp < 1 ? 1 : (probe_spp > 8 ? 8 : probe_spp);
        probe_spp = tune("RTOW_PROBE_SPP", probe_spp, 1, 64);
        if (probe_spp > p->samples_per_pixel) probe_spp = p->samples_per_pixel;
        if (split && !f.pix_cost) {
            HIP_TRY(hipMalloc((void **)&f.pix_cost, (size_t)f.n_pixels * sizeof(uint32_t)));
            HIP_TRY(hipMalloc((void **)&f.heavy_list, (size_t)f.n_pixels * sizeof(uint32_t)));
            HIP_TRY(hipMalloc((void **)&f.heavy_count, 64));
            HIP_TRY(hipMalloc((void **)&f.pix_class, (size_t)f.n_pixels));
            HIP_TRY(hipStreamCreateWithFlags(&f.aux_stream, hipStreamNonBlocking));
            for (int k = 0; k < 2; k++) HIP_TRY(hipEventCreateWithFlags(&f.ev_aux[k], hipEventDisableTiming));
        }
        RenderArgs probe = ra;
        probe.probe = 1;
        probe.spp = probe_spp;
        probe.accum = nullptr;
        probe.spp_before = 0;
        probe.tile_cost = rank_tiles ? f.tile_cost : nullptr;
        probe.tile_order = nullptr;
        probe.pix_cost = split ? f.pix_cost : nullptr;
        if (rank_tiles) HIP_TRY(hipMemsetAsync(f.tile_cost, 0, f.n_tiles * sizeof(uint32_t), stream));
        HIP_TRY(p->variant ? launch_render_fast(ds, probe, stream) : launch_render_strict(ds, probe, stream));
        if (rank_tiles) {
            // the order is kept row-major only where the heaviest tile is within an eighth of the mean (r3: was x4, which sorted for
            // glass only; C3 +1.7 % with every spread sorted, C2 / C5 indifferent between 9 / 8 and 32 / 8, one call)
            const uint32_t flat_x8 = (uint32_t)tune("RTOW_TILE_FLAT_X8", 9, 8, 1 << 20);
            HIP_TRY(launch_tile_order(f.tile_cost, f.tile_order, f.n_tiles, flat_x8, stream));
            ra.tile_order = f.tile_order;
        }
        HIP_TRY(hipMemsetAsync(f.ray_counter, 0, 2 * sizeof(unsigned long long), stream));  // rays, (light) queue cursor
        if (split) {
            // Serving settings (r3; every number below is the mean of several frames per setting in one gpurun call -- earlier sweeps
            // took the best of two runs and missed a bimodal default; profiles/r03_c2_serving_sweep.txt, r03_c3_serving_sweep.txt,
            // r03_rank_serving_sweep.txt, r03_c5_roles.txt).  What per-pixel stamps (RT_STAMP builds, RTOW_PRINT_TAIL) showed: a light
            // pixel just under the threshold runs at a light wave's 30-40 us per ray from the frame's first millisecond to its last,
            // a listed pixel at 7-10 us -- threshold, serving capacity and the longest listed chain have to be moved together.
            //   sphere lists (C2)      two tiers: from 12 rays per sample four pixels to a serving wave (16 lanes per ray), from 9 eight;
            //                          two serving waves of four per workgroup.  Eight to a wave for all: 188...246 ms by which wave
            //                          held the longest chains; one tier of four: 198.6 ms; two tiers: 190.5
            //   primitive BVH (C3)     from 9 rays per sample six to a serving wave, three serving waves of twelve, eight rehearsed
            //                          samples; from 30 rays per sample ONE to a wave (super_list): 162 -> 153 ms
            //   deep segmented (C5)    only for frames of at most seven generations of pixels per lane (deep_roles above), 32 to a
            //                          serving wave: 8 / 12 / 16 rays per sample and 10 / 6 / 4 serving waves of twelve for up to
            //                          2.2 / 5 / 7 generations
            // A frame of few generations of pixels per lane (a rank's stripes) keeps these thresholds -- lower ones helped three
            // ranks of eight and cost the rank with the longest chains a third -- and lets its serving waves take fewer pixels each
            // (adaptive_ppw below).
            const bool few_generations = roles_in_one_launch && generations <= 3.0;  // of pixels per resident lane (twelve waves per CU)
            const int heavy_rays_per_sample = tune("RTOW_HEAVY_RAYS", deep_roles ? (generations <= 2.2 ? 8 : (generations <= 5.0 ? 12 : 16)) : ((sphere_list_kernel && !roles_in_one_launch) ? 12 : 9), 1, 1 << 20);
            int heavy_ppw = tune("RTOW_HEAVY_PPW", deep_roles ? 32 : (sphere_list_kernel ? (roles_in_one_launch ? 8 : 4) : 6), 1, 64);
            const int heavy_blocks = tune("RTOW_HEAVY_BLOCKS", sphere_list_kernel ? f.num_cus / 2 : f.num_cus, 1, 1 << 20);
            // the serving waves' rays are the frame's critical path
            const int heavy_prio = tune("RTOW_HEAVY_PRIO", (sphere_list_kernel && roles_in_one_launch) ? 3 : 0, 0, 3);
            HIP_TRY(hipMemsetAsync(f.heavy_count, 0, 64, stream));
            const int super_rays = tune("RTOW_SUPER_RAYS", (roles_in_one_launch && !deep_roles) ? (sphere_list_kernel ? 12 : 30) : 0, 0, 1 << 20);
            const bool longest = roles_in_one_launch && super_rays > 0;
            if (longest && !f.super_list) HIP_TRY(hipMalloc((void **)&f.super_list, (size_t)f.n_pixels * sizeof(uint32_t)));
            HIP_TRY(launch_classify_pixels(f.pix_cost, f.n_pixels, (uint32_t)(heavy_rays_per_sample * probe_spp), f.pix_class,
                                           f.heavy_list, f.heavy_count, stream, longest ? f.super_list : nullptr, (uint32_t)(super_rays * probe_spp),
                                           (uint32_t)f.width, (uint32_t)tune("RTOW_NEAR_PERCENT", prim_bvh_kernel ? 70 : 0, 0, 100),
                                           (uint32_t)tune("RTOW_NEAR_NEIGHBOURS", prim_bvh_kernel ? 3 : 0, 0, 8)));
            // (primitive BVH worlds: a pixel probed at 70 % of the threshold with three of its eight neighbours over it is listed too --
            // the last pixel of a C3 frame was a light one probed at 8.75 rays per sample in a patch of heavy ones, really costing 16:
            // 152.5 -> 147.8 ms, six frames per setting; sphere lists: no difference, left off)
            if (longest) {
                HIP_TRY(hipMemsetAsync(f.ray_counter + 9, 0, sizeof(unsigned long long), stream));
                ra.super_list = f.super_list;
                ra.super_count = f.heavy_count + 1;
                ra.super_cursor = reinterpret_cast<uint32_t *>(f.ray_counter + 9);
                ra.super_ppw = tune("RTOW_SUPER_PPW", sphere_list_kernel ? 4 : 1, 1, 64);
            }
            HIP_TRY(hipMemsetAsync(f.ray_counter + 6, 0, sizeof(unsigned long long), stream));  // heavy queue cursor
            if (roles_in_one_launch) {
                if (sphere_list_kernel && ra.max_blocks_per_cu <= 0) ra.max_blocks_per_cu = 3;
                ra.heavy_list = f.heavy_list;
                ra.heavy_count = f.heavy_count;
                ra.heavy_cursor = reinterpret_cast<uint32_t *>(f.ray_counter + 6);
                ra.heavy_waves = tune("RTOW_HEAVY_WAVES", deep_roles ? (generations <= 2.2 ? 10 : (generations <= 5.0 ? 6 : 4)) : (sphere_list_kernel ? 2 : 3), 0, 12);
                ra.heavy_ppw = heavy_ppw;
                ra.heavy_priority = heavy_prio;
                // fewer pixels per serving wave than the tuned numbers where the light pixels are few -- up to three generations of
                // pixels per lane, i.e. a rank's stripes of a split frame: the light side is short there and a listed chain is
                // shortest with its wave to itself (slowest rank, C2 / 4: 132 -> 119 ms, / 8: 135 -> 97; C3 / 2: 153 -> 135, / 4:
                // 154 -> 124, / 8: 155 -> 122).  A full frame packs the serving waves as densely as tuned: the ones left over join
                // the light queue at once (C3, 4.9 generations: 152 against 159 ms).
                ra.adaptive_ppw = tune("RTOW_ADAPTIVE_PPW", few_generations ? 1 : 0, 0, 1);
                ra.pix_class = f.pix_class;
            } else {
                HIP_TRY(hipEventRecord(f.ev_aux[0], stream));
                HIP_TRY(hipStreamWaitEvent(f.aux_stream, f.ev_aux[0], 0));
                RenderArgs heavy = ra;
                heavy.tile_order = nullptr;
                heavy.pixel_list = f.heavy_list;
                heavy.pixel_list_count = f.heavy_count;
                heavy.cursor = reinterpret_cast<uint32_t *>(f.ray_counter + 6);
                heavy.pixels_per_wave = heavy_ppw;
                if (sphere_list_kernel) heavy.coop_threshold = 65;  // always the grouped scan
                heavy.grid_blocks = heavy_blocks;
                heavy.max_blocks_per_cu = 8;
                heavy.wave_priority = heavy_prio;
                HIP_TRY(p->variant ? launch_render_fast(ds, heavy, f.aux_stream) : launch_render_strict(ds, heavy, f.aux_stream));
                HIP_TRY(hipEventRecord(f.ev_aux[1], f.aux_stream));
                ra.pix_class = f.pix_class;
            }
        }
    }
    HIP_TRY(p->variant ? launch_render_fast(ds, ra, stream) : launch_render_strict(ds, ra, stream));
    if (split && !roles_in_one_launch) HIP_TRY(hipStreamWaitEvent(stream, f.ev_aux[1], 0));
    HIP_TRY(hipEventRecord(f.ev[2], stream));
    // The counters come home on the film's own stream: a blocking hipMemcpy in rt_render_finish would wait for every
    // other film's frame as well and serialise frames that were launched to overlap.
    HIP_TRY(hipMemcpyAsync(f.host_counters, f.ray_counter, kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipEventRecord(f.ev[3], stream));
    return RT_OK;
}

int rt_render_launch(rt_scene *scene, rt_film *film, const rt_render_params *p)
{
    if (!scene || !film || !p) return fail(RT_ERR_INVALID, "rt_render_launch: null argument");
    SceneImpl &s = *S(scene);
    FilmImpl &f = *F(film);
    if (p->width != f.width || p->height != f.height || p->stripe_rows != f.stripe_rows || p->rank != f.rank ||
        p->world_size != f.world_size || p->device != f.device)
        return fail(RT_ERR_INVALID, "rt_render_launch: params do not match the film's geometry/device");
    if (p->samples_per_pixel < 0 || p->max_depth < 0) return fail(RT_ERR_INVALID, "rt_render_launch: negative spp/depth");
    if (p->variant != 0 && p->variant != 1) return fail(RT_ERR_INVALID, "rt_render_launch: variant must be 0 (strict) or 1 (fast)");
    if (p->pixels_per_wave < 0 || p->pixels_per_wave > 64) return fail(RT_ERR_INVALID, "rt_render_launch: pixels_per_wave must be 0 (automatic) or 1..64");
    if (f.in_flight)
        return fail(RT_ERR_STATE, "rt_render_launch: this film already has a render in flight (rt_render_finish it first; "
                                  "use one film per frame in flight)");
    if (int rc = rt_scene_upload(scene, f.device)) return rc;
    if (int rc = select_device(f.device)) return rc;
    hipStream_t stream = p->stream ? (hipStream_t)p->stream : f.own_stream;
    // In flight from here on: the scene may not change (or go away) under a kernel that is already on the stream, also
    // when a later step of the launch fails.
    f.last_stream = stream;
    mark_in_flight(s, f);
    const int rc = enqueue_frame(s, f, p, stream);
    if (rc != RT_OK) {
        const std::string why = rt_last_error();
        hipStreamSynchronize(stream);  // whatever did get enqueued
        mark_done(f);
        set_error(why);
        return rc;
    }
    f.last_samples = (uint64_t)f.n_pixels * (uint64_t)p->samples_per_pixel;
    f.last_variant = p->variant;
    return RT_OK;
}

int rt_render_finish(rt_scene *scene, rt_film *film, rt_render_stats *stats)
{
    (void)scene;
    if (!film) return fail(RT_ERR_INVALID, "rt_render_finish: null film");
    FilmImpl &f = *F(film);
    if (!f.in_flight) return fail(RT_ERR_STATE, "rt_render_finish: nothing launched");
    if (int rc = select_device(f.device)) return rc;
    const hipError_t waited = hipEventSynchronize(f.ev[3]);
    if (waited != hipSuccess && f.last_stream) hipStreamSynchronize(f.last_stream);  // whatever is left on the stream
    mark_done(f);  // also when the wait failed: the scene must not stay locked for ever
    if (waited != hipSuccess) return hip_fail(waited, "hipEventSynchronize(render finished)");
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        float ms_seed = 0, ms_render = 0;
        HIP_TRY(hipEventElapsedTime(&ms_seed, f.ev[0], f.ev[1]));
        HIP_TRY(hipEventElapsedTime(&ms_render, f.ev[1], f.ev[2]));
        const unsigned long long rays = f.host_counters[0];
        if (tune_set("RTOW_PRINT_HEAVY") && f.heavy_count && f.pix_cost) {  // diagnostics: the rehearsal's cost classes
            uint32_t n_heavy = 0;
            std::vector<uint32_t> cost(f.n_pixels);
            HIP_TRY(hipMemcpy(&n_heavy, f.heavy_count, sizeof n_heavy, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(cost.data(), f.pix_cost, cost.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            unsigned long long hist[16] = {};
            uint32_t top = 0;
            for (uint32_t c : cost) {
                hist[c / 16 > 15 ? 15 : c / 16]++;
                top = c > top ? c : top;
            }
            std::fprintf(stderr, "heavy pixels %u of %u; probe rays per pixel: max %u; histogram by 16:", n_heavy, f.n_pixels, top);
            for (int k = 0; k < 16; k++) std::fprintf(stderr, " %llu", hist[k]);
            std::fprintf(stderr, "\n");
        }
        if (tune_set("RTOW_PRINT_TAIL") && f.dbg_times) {  // who finishes last: the 24 last pixels and a histogram of the ends
            std::vector<uint32_t> t((size_t)f.n_pixels * 2), cost(f.n_pixels, 0);
            HIP_TRY(hipMemcpy(t.data(), f.dbg_times, t.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            if (f.pix_cost) HIP_TRY(hipMemcpy(cost.data(), f.pix_cost, cost.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            const uint32_t t0 = (uint32_t)f.host_counters[5];
            std::vector<uint32_t> order(f.n_pixels);
            for (uint32_t k = 0; k < f.n_pixels; k++) order[k] = k;
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (uint32_t)(t[2 * a + 1] - t0) > (uint32_t)(t[2 * b + 1] - t0); });
            const double last = (uint32_t)(t[2 * order[0] + 1] - t0) * 1e-5;
            unsigned long long hist[12] = {};
            double mean_dur[12] = {}, mean_cost[12] = {};
            for (uint32_t k = 0; k < f.n_pixels; k++) {
                const double end = (uint32_t)(t[2 * k + 1] - t0) * 1e-5, dur = (uint32_t)(t[2 * k + 1] - t[2 * k]) * 1e-5;
                int b = (int)((last - end) / 5.0);
                b = b > 11 ? 11 : b;
                hist[b]++; mean_dur[b] += dur; mean_cost[b] += cost[k];
            }
            std::fprintf(stderr, "tail: last pixel ends +%.1f ms; pixels ending in the last 5 ms steps (count, mean duration ms, mean probe rays):", last);
            for (int b = 0; b < 12; b++) std::fprintf(stderr, " [%llu %.1f %.1f]", hist[b], hist[b] ? mean_dur[b] / hist[b] : 0.0, hist[b] ? mean_cost[b] / hist[b] : 0.0);
            std::fprintf(stderr, "\n");
            for (int k = 0; k < 24; k++) {
                const uint32_t px = order[k * 40];
                std::fprintf(stderr, "  pixel row %u col %u: start +%.1f end +%.1f ms, probe rays %u\n", px / (uint32_t)f.width, px % (uint32_t)f.width,
                             (uint32_t)(t[2 * px] - t0) * 1e-5, (uint32_t)(t[2 * px + 1] - t0) * 1e-5, cost[px]);
            }
        }
        if (tune_set("RTOW_PRINT_STAMPS")) {  // diagnostic builds (-DRT_STAMP=1): 100 MHz wall-clock ticks
            const unsigned long long *st = f.host_counters;
            std::fprintf(stderr, "stamps: start %llu  queue exhausted +%.3f ms  first wave out +%.3f ms  last wave out +%.3f ms  heavy pixels done +%.3f ms\n",
                         st[5], (st[2] - st[5]) * 1e-5, (st[4] - st[5]) * 1e-5, (st[3] - st[5]) * 1e-5, st[8] ? (st[8] - st[5]) * 1e-5 : 0.0);
        }
        if (tune_set("RTOW_PRINT_PHASES")) {  // diagnostic builds (-DRT_PHASES=1)
            const unsigned long long *c = f.host_counters;
            const char *name[24] = {"node step", "leaf test", "shade", "refill", "  group/instance", "  medium", "  primitive", "",
                                    "    record+xforms", "    box", "    sub-BVH", "    other geometry", "between walks again", "limited node pass", "node visits (lanes)", "",
                                    "box pass", "medium pass / between walks", "object pass", "primitive pass", "  hit record", "  scatter",
                                    "  next camera ray", "  pixel done"};
            if ((f.last_kernel.kind & 63) >= 16) {  // sphere-list kernel: slots 0 / 1 are its two scans
                name[0] = "scan, pixel-parallel";
                name[1] = "scan, cooperative";
            }
            const double total = (double)c[7];
            for (int k = 0; k < 24; k++)
                if (name[k][0] && c[96 + k])
                    std::fprintf(stderr, "phase %-20s: %5.1f %% of wave time, %10llu passes, %5.1f lanes/pass, %7.0f cycles/pass\n", name[k],
                                 100.0 * c[32 + k] / total, c[96 + k], (double)c[64 + k] / c[96 + k], (double)c[32 + k] / c[96 + k]);
        }
        stats->samples = f.last_samples;
        stats->rays = rays;
        stats->seconds_seed = ms_seed * 1e-3;
        stats->seconds_render = ms_render * 1e-3;
        stats->pixels = f.n_pixels;
        stats->rows = (uint32_t)f.rows_owned;
        stats->kernel_vgprs = (uint32_t)f.last_kernel.vgprs;
        stats->lds_bytes = (uint32_t)f.last_kernel.lds_bytes;
        stats->kernel_kind = (uint32_t)f.last_kernel.kind;
        stats->pixels_per_wave = (uint32_t)f.last_pixels_per_wave;
    }
    return RT_OK;
}

int rt_film_download(rt_film *film, double *frame_full, int width, int height)
{
    if (!film || !frame_full) return fail(RT_ERR_INVALID, "rt_film_download: null argument");
    FilmImpl &f = *F(film);
    if (width != f.width || height != f.height) return fail(RT_ERR_INVALID, "rt_film_download: frame size mismatch");
    if (int rc = select_device(f.device)) return rc;
    std::vector<double> compact((size_t)f.n_pixels * 3);
    if (f.n_pixels) HIP_TRY(hipMemcpy(compact.data(), f.pixels, compact.size() * sizeof(double), hipMemcpyDeviceToHost));
    size_t lr = 0;
    for (int j = 0; j < height; j++)
        if ((j / f.stripe_rows) % f.world_size == f.rank) {
            std::memcpy(frame_full + (size_t)j * width * 3, compact.data() + lr * (size_t)width * 3, sizeof(double) * (size_t)width * 3);
            lr++;
        }
    return RT_OK;
}

int rt_render(rt_scene *scene, const rt_render_params *params, double *frame, rt_render_stats *stats)
{
    if (!scene || !params || !frame) return fail(RT_ERR_INVALID, "rt_render: null argument");
    rt_film *film = rt_film_create(params->device, params->width, params->height, params->stripe_rows > 0 ? params->stripe_rows : 8,
                                   params->rank, params->world_size > 0 ? params->world_size : 1);
    if (!film) return std::strstr(rt_last_error(), "no HIP device") ? RT_ERR_NO_DEVICE : RT_ERR_HIP;
    rt_render_params p = *params;
    if (p.stripe_rows <= 0) p.stripe_rows = 8;
    if (p.world_size <= 0) p.world_size = 1;
    int rc = rt_render_launch(scene, film, &p);
    if (rc == RT_OK) rc = rt_render_finish(scene, film, stats);
    if (rc == RT_OK) rc = rt_film_download(film, frame, p.width, p.height);
    rt_film_destroy(film);
    return rc;
}

} // extern "C"
