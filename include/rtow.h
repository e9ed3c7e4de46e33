/*
 * rtow.h -- C-ABI of the MI355X-native path tracer (librtow_hip.so).
 *
 * The reference (eazuooz/RayTracinginOneWeekendinCUDA) has no FFI layer; its hot path sits behind two
 * internal C++ interfaces.  This header is the drop-in boundary for both:
 *
 *   1. the CONSTRUCTION API that CreateWorld calls (R/kernel.cu:176-543): one extern "C" function per
 *      reference constructor, same parameter order and meaning, returning a handle instead of a
 *      device pointer;
 *   2. the RENDER API that main() calls (R/kernel.cu:570-742): RenderInit + Render launches, the
 *      framebuffer hand-back and the PPM writer.
 *
 * R/ = /root/reference/RayTracinginOneWeekend/.  Plain pointers and sizes only; no C++ or torch types.
 * Every function returning int returns 0 on success and a non-zero rt_status otherwise; functions
 * returning rt_handle return 0 on error.  rt_last_error() gives the message of the calling thread's
 * last failure.  A scene is thread-compatible (one thread at a time), not thread-safe.
 */
#ifndef RTOW_H
#define RTOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTOW_API __attribute__((visibility("default")))

typedef uint32_t rt_handle;             /* 0 = invalid */
typedef struct rt_scene rt_scene;       /* owns every object built through it (host arena + device tables) */
typedef struct rt_rng rt_rng;           /* host-side curandState replacement for scene generation */
typedef struct rt_film rt_film;         /* framebuffer + per-pixel RNG state on one GPU */

enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = 1,        /* bad argument / handle */
    RT_ERR_UNSUPPORTED = 2,    /* object nesting the flattener cannot normalise */
    RT_ERR_HIP = 3,            /* a HIP call failed (message carries the hipError_t) */
    RT_ERR_NO_DEVICE = 4,      /* no usable gfx950 device */
    RT_ERR_STATE = 5           /* call order violated (e.g. render before commit) */
};

RTOW_API const char *rt_last_error(void);
RTOW_API const char *rt_version(void);

/* ---- RNG: curand_init(seed, sequence, 0) / curand_uniform (R/kernel.cu:101-107, RND macro :157) ---- */
RTOW_API rt_rng *rt_rng_create(uint64_t seed, uint64_t sequence);
RTOW_API void rt_rng_destroy(rt_rng *rng);
RTOW_API float rt_rng_uniform(rt_rng *rng);                       /* float in (0,1] */
RTOW_API uint32_t rt_rng_next_u32(rt_rng *rng);
RTOW_API void rt_rng_state(const rt_rng *rng, uint32_t out6[6]);   /* {d, v0..v4} */
/* salt_kind 0 = cuRAND device-API seed salts (the product's RNG); 1 = rocRAND's salts (test cross-check only) */
RTOW_API rt_rng *rt_rng_create_salted(uint64_t seed, uint64_t sequence, int salt_kind);

/* ---- scene lifetime ---- */
RTOW_API rt_scene *rt_scene_create(void);
RTOW_API void rt_scene_destroy(rt_scene *scene);                   /* replaces FreeWorld, R/kernel.cu:548-568 */
/* Destroying a scene while launches of it are in flight is allowed: the call first waits for every such launch (the
 * kernels read the scene's tables); the films stay valid and their rt_render_finish reports as usual. */
/* Options read by the next rt_scene_commit (tests and timing; none changes an image). */
#define RT_SCENE_PLAIN_QUADS 1u          /* every quad takes the general test of R/Quad.h:52-99 (default: quads along the coordinate
                                            axes drop the terms that are exact zeros; MakeBox boxes are tested as boxes) */
#define RT_SCENE_REFERENCE_TREE_ONLY 2u  /* do not build the library's own tree for primitive worlds: only the reference's */
RTOW_API int rt_scene_set_options(rt_scene *scene, uint32_t options);

/* ---- textures (R/Texture.h) ---- */
RTOW_API rt_handle rt_solid_color(rt_scene *s, double r, double g, double b);                 /* :38,:43 */
RTOW_API rt_handle rt_checker_texture(rt_scene *s, double scale, rt_handle even, rt_handle odd); /* :63 */
/* bytes are copied; w*h*3 RGB, row 0 = top (what RtwImage hands to ImageTexture, R/RtwImage.h:51-92). NULL data => cyan. */
RTOW_API rt_handle rt_image_texture(rt_scene *s, const unsigned char *rgb, int width, int height); /* :103 */
RTOW_API rt_handle rt_noise_texture(rt_scene *s, double scale, rt_rng *rng);                  /* :153; draws from rng */
/* What RtwImage::Load does to decoded 8-bit pixels before ImageTexture sees them (R/RtwImage.h:54,66-67,100-105 on top
 * of stbi_loadf's LDR->HDR step, R/external/stb_image.h:1869): out = FloatToByte((float)pow(in / 255.0f, 2.2f)).
 * For decoded pixels that come from elsewhere (another decoder's output differs from stb's by up to 3 in ~0.6 % of the bytes of
 * the reference's earthmap.jpg); rt_rtwimage_load below does the whole of RtwImage::Load.  in/out may alias. */
RTOW_API void rt_rtwimage_bytes(const unsigned char *decoded_srgb, size_t count, unsigned char *out);
/* RtwImage::Load itself (R/RtwImage.h:51-87 over stbi_loadf, R/StbImageImpl.cpp): read a JPEG file and return the width * height * 3
 * bytes the reference hands to ImageTexture (row 0 = top), bit for bit what the reference's stb_image build produces for a
 * sequential Huffman JPEG of 8-bit samples, grey or YCbCr / RGB, any sampling factors, with or without restart intervals
 * (csrc/jpeg_decode.cpp restates that decoder's inverse DCT, upsampling and colour conversion).  Progressive / arithmetic-coded /
 * 12-bit / CMYK files return RT_ERR_UNSUPPORTED -- ImageTexture(NULL) then renders the reference's cyan fallback.
 * rt_jpeg_decode: the 8-bit sRGB pixels only (stbi_load), without RtwImage's linearisation.  Free the result with rt_image_free. */
RTOW_API int rt_rtwimage_load(const char *path, unsigned char **rgb_out, int *width, int *height);
RTOW_API int rt_jpeg_decode(const unsigned char *data, size_t size, unsigned char **rgb_out, int *width, int *height);
RTOW_API void rt_image_free(unsigned char *rgb);

/* ---- materials (R/Material.h, R/Metal.h, R/Dielectric.h) ---- */
RTOW_API rt_handle rt_lambertian(rt_scene *s, double r, double g, double b);                  /* Material.h:57 */
RTOW_API rt_handle rt_lambertian_tex(rt_scene *s, rt_handle texture);                         /* Material.h:63 */
RTOW_API rt_handle rt_metal(rt_scene *s, double r, double g, double b, double fuzz);          /* Metal.h:12 */
RTOW_API rt_handle rt_dielectric(rt_scene *s, double refraction_index);                       /* Dielectric.h:13 */
RTOW_API rt_handle rt_diffuse_light(rt_scene *s, double r, double g, double b);               /* Material.h:109 */
RTOW_API rt_handle rt_diffuse_light_tex(rt_scene *s, rt_handle texture);                      /* Material.h:103 */
RTOW_API rt_handle rt_isotropic(rt_scene *s, double r, double g, double b);                   /* Material.h:142 */
RTOW_API rt_handle rt_isotropic_tex(rt_scene *s, rt_handle texture);                          /* Material.h:147 */

/* ---- hittables ---- */
RTOW_API rt_handle rt_sphere(rt_scene *s, double cx, double cy, double cz, double radius, rt_handle material); /* Sphere.h:12 */
RTOW_API rt_handle rt_moving_sphere(rt_scene *s, double c0x, double c0y, double c0z, double c1x, double c1y,
                                    double c1z, double time0, double time1, double radius,
                                    rt_handle material);                                       /* MovingSphere.h:19 */
RTOW_API rt_handle rt_quad(rt_scene *s, const double q[3], const double u[3], const double v[3],
                           rt_handle material);                                                /* Quad.h:25 */
RTOW_API rt_handle rt_translate(rt_scene *s, rt_handle object, double ox, double oy, double oz); /* Instance.h:31 */
RTOW_API rt_handle rt_rotate_y(rt_scene *s, rt_handle object, double angle_degrees);          /* Instance.h:74 */
RTOW_API rt_handle rt_make_box(rt_scene *s, const double a[3], const double b[3], rt_handle material); /* Instance.h:166 */
RTOW_API rt_handle rt_hittable_list(rt_scene *s, const rt_handle *objects, int count);        /* HittableList.h:21 */
RTOW_API rt_handle rt_constant_medium(rt_scene *s, rt_handle boundary, double density, double r, double g,
                                      double b);                                               /* ConstantMedium.h:39 */
RTOW_API rt_handle rt_constant_medium_tex(rt_scene *s, rt_handle boundary, double density,
                                          rt_handle texture);                                  /* ConstantMedium.h:32 */
/* BvhNode(objects, 0, count, ...): builds the tree with the reference's rule and PERMUTES objects[] in
 * place exactly as the reference's DeviceSort does (R/BvhNode.h:50-90,180-193). */
RTOW_API rt_handle rt_bvh_node(rt_scene *s, rt_handle *objects, int count);                   /* BvhNode.h:50 */
RTOW_API int rt_hittable_bounding_box(rt_scene *s, rt_handle object, double out_xyz_minmax[6]); /* Hittable.h:60 */

/* ---- world + camera (the two outputs of CreateWorld, R/kernel.cu:523-541) ---- */
RTOW_API int rt_scene_set_world(rt_scene *s, rt_handle world);      /* a BvhNode, a HittableList or any hittable */
RTOW_API int rt_scene_set_camera(rt_scene *s, const double lookfrom[3], const double lookat[3],
                                 const double vup[3], double vfov_degrees, double aspect, double aperture,
                                 double focus_dist, double time0, double time1,
                                 const double background[3]);                                  /* Camera.h:36-72 */

/* Built-in scenes: ids 0..9 = the reference's sceneId (R/kernel.cu:199-517); 10 = three-spheres (config C1);
 * 11 = scene 0 with every MovingSphere made static (config C2).  world_kind 0 = BvhNode world (the
 * reference's), 1 = HittableList world ("no BVH").  earth_rgb may be NULL (scenes 2 and 9 then show cyan). */
RTOW_API int rt_scene_build_builtin(rt_scene *s, int scene_id, int world_kind, int image_width,
                                    int image_height, uint64_t seed, const unsigned char *earth_rgb,
                                    int earth_w, int earth_h);

/* Flatten the object graph into the SoA tables the kernel reads (host only; no GPU needed). */
RTOW_API int rt_scene_commit(rt_scene *s);

typedef struct rt_scene_info {
    uint32_t world_kind;          /* 0 bvh, 1 list */
    uint32_t n_leaves;            /* top-level leaves of the world */
    uint32_t n_nodes;             /* threaded BVH nodes */
    uint32_t n_spheres, n_moving_spheres, n_quads;
    uint32_t n_objects;           /* composite leaves (instances, boxes, media) */
    uint32_t n_xforms, n_media, n_materials, n_textures, n_perlin, n_images;
    uint32_t table_bytes;         /* bytes of the geometry tables staged on chip */
    uint32_t image_bytes;
    uint32_t reserved[3];
} rt_scene_info;
RTOW_API int rt_scene_get_info(rt_scene *s, rt_scene_info *out);

/* Introspection for tests (valid after commit): world leaves in final order. kind: 0 sphere, 1 moving
 * sphere, 2 quad, 3 composite object; box = {xmin,xmax,ymin,ymax,zmin,zmax}. Returns the leaf count. */
RTOW_API int rt_scene_dump_leaves(rt_scene *s, int max_leaves, int *kind_out, double *box_out);
/* Threaded-BVH nodes in preorder: box[6], a, b, escape per node (a,b = leaf refs or 0xE0000000 for inner). */
RTOW_API int rt_scene_dump_nodes(rt_scene *s, int max_nodes, double *box_out, uint32_t *abe_out);
/* The library's own tree for primitive-only BVH worlds (0 nodes if the world has none): box[6], a, b, and per direction
 * octant the {hit, escape} links (16 x uint16 per node, 0xFFFF = end). */
RTOW_API int rt_scene_dump_fast_nodes(rt_scene *s, int max_nodes, double *box_out, uint32_t *ab_out, uint16_t *link_out);
RTOW_API int rt_scene_dump_camera(rt_scene *s, double out27[27]);

/* ---- render (RenderInit + Render, R/kernel.cu:110-154,675-691) ---- */
typedef struct rt_render_params {
    int32_t width, height;        /* full frame (maxX, maxY) */
    int32_t samples_per_pixel;    /* numSamples */
    int32_t max_depth;            /* literal 50 at R/kernel.cu:71 */
    uint64_t seed;                /* literal 1984 at R/kernel.cu:118 */
    int32_t stripe_rows;          /* multi-GPU: rows are dealt in stripes of this many rows ... */
    int32_t rank, world_size;     /* ... stripe k belongs to rank k % world_size.  1 GPU: rank 0 of 1 */
    int32_t variant;              /* 0 = strict: no FMA contraction, the reference's arithmetic operation by operation; frames equal the CPU
                                     oracle's bit for bit -- the build to use where the north star's 1e-5 tolerance matters.
                                     1 = fast: FMA contraction, 0-3 % faster.  A contracted comparison that falls the other way re-draws
                                     the rest of that pixel's random stream: rare at a few hundred samples on surface scenes (C2-C4:
                                     >= 0.9998 of the pixels within 1e-5), but on media scenes at thousands of samples per pixel most
                                     pixels leave the tolerance (C5 at 5000 spp: 0.24-0.75 within 1e-5; still a correct image) */
    int32_t device;               /* HIP device ordinal */
    int32_t flags;                /* RT_FLAG_* */
    void *stream;                 /* hipStream_t to launch on (NULL = the film's own stream) */
    int32_t coop_threshold;       /* tuning: sphere-list waves with fewer live lanes scan cooperatively (0 = default) */
    int32_t overdue_rays_per_sample; /* tuning: a pixel past this many rays/sample advances in extra cooperative passes (0 or <0 = never, the default) */
    int32_t shade_batch;          /* tuning: BVH kernels shade once this many lanes finished traversal (0 = default 16) */
    int32_t max_blocks_per_cu;    /* tuning: cap on resident 256-thread workgroups per CU (0 = as many as fit) */
    int32_t pixels_per_wave;      /* list worlds (HittableList worlds and small BVH worlds rendered as lists; no media): pixels a wave works
                                     on at a time, a power of two 1..64; the wave's other lanes share each ray's leaf tests (64 / pixels
                                     lanes per ray), the same frame bit for bit.  0 or 64 = one lane per ray, which is also the fastest
                                     setting for every frame size measured (a mixed list's leaves are different
                                     code, which a group of lanes runs one after the other like a single lane does); smaller values are
                                     for lists of one kind of leaf and for experiments */
    int32_t reserved0;
} rt_render_params;

#define RT_FLAG_KEEP_RNG_STATE 1u  /* do not re-seed: continue from the film's saved per-pixel state (progressive) */
#define RT_FLAG_OVERDUE_PRIORITY 4u /* tuning/diagnostics: overdue pixels raise their wave's priority instead of going cooperative */
#define RT_FLAG_ACCUMULATE 8u      /* progressive: with KEEP_RNG_STATE, add this launch's samples to the film's running sums;
                                      the pixels then hold sqrt(sum / all samples so far), bit-identical to one launch of that many spp */
#define RT_FLAG_FORCE_GENERAL 2u   /* tests: run the general kernel even where a specialised instantiation applies */
#define RT_FLAG_ALWAYS_WALK 32u     /* small BVH worlds without media (up to 16 cheap leaves) are rendered by scanning all leaves in the tree's
                                      leaf order (same closest hit, no node visits); this flag walks the tree anyway */
#define RT_FLAG_NO_PIXEL_CLASSES 64u /* hand every pixel out through the one tile queue (default: a rehearsal of the first samples lists the
                                      pixels with long ray chains, and some waves of every workgroup serve those first, a few pixels per
                                      wave, before they join the tile queue, which skips them -- sphere lists in two tiers of 4 and 8
                                      pixels per wave with the lanes sharing each ray's scan, primitive BVH worlds on the library's tree
                                      with the very longest chains one to a wave, deep composite worlds only where the frame is a few
                                      generations of pixels on the GPU's lanes, e.g. one rank's stripes; the image is the same either way) */
#define RT_FLAG_REFERENCE_TREE 128u  /* BVH worlds of primitives only are walked through the library's own tree (surface-area heuristic, near
                                      child first) -- no leaf draws random numbers there, so the closest hit is the one the reference's tree
                                      gives; this flag walks the reference's own tree in its own order instead (tests, timing).  A world in
                                      which two primitives coincide (identical spheres, overlapping quads in one plane) has no library
                                      tree at all: there the order of the tests decides which of the two a ray sees */
#define RT_FLAG_EXACT_SCAN 256u      /* sphere-list worlds: every ray runs the reference's discriminant against every sphere (default: a cheaper
                                      conservative filter rejects the spheres a ray's line misses and only the survivors go through the
                                      reference's arithmetic; the image is the same bit for bit either way) */
#define RT_FLAG_FILTER_FP64 2048u     /* sphere-list worlds: the conservative filter in fp64, one sphere per 8 instructions (default: its packed
                                      fp32 form, two spheres per 9 instructions, a little coarser; the survivors always go through the
                                      reference's fp64 test, so the image is the same bit for bit) -- tests, timing */
#define RT_FLAG_ACCELERATE_LISTS 512u /* HittableList worlds of primitives only (no leaf draws random numbers): render through the library's
                                      own tree as a BvhNode world would be -- the reference's "BVH image == list image" invariant the other
                                      way round; off by default so that a list world is scanned as the reference scans it.  Ignored for
                                      a list with coincident primitives (see RT_FLAG_REFERENCE_TREE) */
#define RT_FLAG_COOP_SINGLE 1024u    /* tests: sphere-list worlds, thin waves scan one ray at a time with all 64 lanes (the older scheme)
                                      instead of several rays in groups of lanes */
#define RT_FLAG_ROW_MAJOR_TILES 16u /* BVH worlds: keep the pixel queue in row-major tile order (default: a short rehearsal ranks
                                      the 8x8 tiles by rays traced and the heaviest start first; the image is the same either way) */

typedef struct rt_render_stats {
    uint64_t samples;             /* pixels rendered by this rank x spp */
    uint64_t rays;                /* RayColor loop iterations (one world Hit each) */
    double seconds_seed;          /* RNG seeding kernel, HIP events */
    double seconds_render;        /* render kernel, HIP events */
    uint32_t pixels;              /* pixels owned by this rank */
    uint32_t rows;                /* rows owned by this rank */
    uint32_t kernel_vgprs;
    uint32_t lds_bytes;
    uint32_t kernel_kind;         /* which instantiation ran: world*4 + composite*2 + rich (world 0 bvh, 1 list, 2 sphere list) */
    uint32_t pixels_per_wave;     /* what rt_render_params.pixels_per_wave came to for this launch (64 = one lane per ray) */
} rt_render_stats;

/* Rows owned by (rank, world_size) for a height: returns count, fills rows_out (ascending j) if non-NULL. */
RTOW_API int rt_stripe_rows(int height, int stripe_rows, int rank, int world_size, int *rows_out, int max_rows);

RTOW_API rt_film *rt_film_create(int device, int width, int height, int stripe_rows, int rank, int world_size);
RTOW_API void rt_film_destroy(rt_film *film);
/* Device pointer of this rank's compact framebuffer: rows_owned x width x 3 doubles (sqrt-gamma applied, like
 * frameBuffer[] at R/kernel.cu:150-153), rows in ascending j. */
RTOW_API void *rt_film_device_pixels(rt_film *film);
/* Render into caller-owned device memory instead (e.g. a torch tensor that RCCL will gather from); must hold
 * rt_film_pixel_bytes() bytes on the film's device.  NULL restores the film's own buffer. */
RTOW_API int rt_film_bind_pixels(rt_film *film, void *device_pixels);
RTOW_API size_t rt_film_pixel_bytes(rt_film *film);

/* Upload the committed scene to a device (idempotent per device). */
RTOW_API int rt_scene_upload(rt_scene *s, int device);

/* Asynchronous on params->stream: seeds (unless KEEP_RNG_STATE) and renders this rank's rows into the film. */
RTOW_API int rt_render_launch(rt_scene *s, rt_film *film, const rt_render_params *params);
/* Waits for the launch, fills stats (HIP-event kernel durations, ray counter). */
RTOW_API int rt_render_finish(rt_scene *s, rt_film *film, rt_render_stats *stats);
/* Copies this rank's compact rows into a full W x H x 3 host frame (pixel (i,j) at (j*W+i)*3, j = 0 bottom). */
RTOW_API int rt_film_download(rt_film *film, double *frame_full, int width, int height);
/* Scatter compact rank buffers (as gathered over RCCL, rank-major) into a full frame; pure host code. */
RTOW_API int rt_deinterleave(const double *gathered, int width, int height, int stripe_rows, int world_size,
                             size_t rank_stride_doubles, double *frame_full);

/* Convenience: create film, upload, render 1 GPU, download.  frame = W*H*3 doubles. */
RTOW_API int rt_render(rt_scene *s, const rt_render_params *params, double *frame, rt_render_stats *stats);

/* Extra outputs: binary PPM (P6, same quantisation as the P3 writer) and PFM (little-endian float32, the
 * gamma-corrected values unclamped, bottom row first as PFM prescribes). */
RTOW_API int rt_write_ppm_binary(const char *path, const double *frame, int width, int height);
RTOW_API int rt_write_pfm(const char *path, const double *frame, int width, int height);

/* PPM writer, byte-for-byte the reference's (R/kernel.cu:696-721): P3, rows from j=H-1 down, clamp [0,0.999], int(256*c). */
RTOW_API int rt_write_ppm(const char *path, const double *frame, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* RTOW_H */
