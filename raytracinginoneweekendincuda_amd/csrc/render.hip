// render.hip -- the gfx950 path-tracing megakernel and the RNG seeding kernel.
//
// Replaces the reference's RenderInit + Render kernels (R/kernel.cu:110-154) and everything they
// inline: Camera::GetRay (R/Camera.h:76-85), RayColor (R/kernel.cu:65-98), BvhNode::Hit
// (R/BvhNode.h:101-158), the primitive / instance / medium Hit functions, Material::Scatter/Emitted
// and Texture::Value.  Not a translation: one lane owns one pixel and runs a flat
// "one ray segment per iteration" loop with path regeneration (a lane whose path ends starts its
// pixel's next sample in the same iteration), virtual dispatch is tag dispatch over the SoA tables
// of flat_scene.h, the BVH is walked stacklessly through escape links with its nodes staged in LDS,
// a list-of-spheres world is scanned with wave-uniform (scalar-path) primitive rows while the rare
// sqrt/divide root work is deferred to a per-lane LDS queue, hit records are built once per bounce
// from (t, primitive) instead of on every accepted candidate, and sphere UVs are computed only when
// an image texture will read them.  Each of these is result-preserving: see DESIGN.md.
//
// This file is compiled twice: RT_STRICT=1 with -ffp-contract=off (no FMA; bit-comparable with the
// CPU oracle) and RT_STRICT=0 with the default contraction (fast variant) -- and each of those in two groups of
// instantiations (RT_GROUP), because they want different code generation: group 0 (sphere-list and primitive-BVH
// kernels, seeding, dispatch) with the compiler's defaults, group 1 (every kernel with composite leaves, media or table
// textures) with machine-level loop-invariant code motion off.  There LICM hoists the libm polynomial coefficients and
// other constants of rarely executed code (log, sin, acos, atan2) out of the main loop into ~100 VGPRs and the register
// allocator then spills them: 520 B/lane of scratch in the deep general kernel, 80 B without (C5 +10 %); the tight
// primitive walk loses 5-7 % the same way, hence the split.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstddef>
#include <cstdint>

#include "flat_scene.h"
#include "render_iface.h"
#include "rng.h"

#ifndef RT_STRICT
#define RT_STRICT 1
#endif
#ifndef RT_GROUP
#define RT_GROUP 0
#endif

namespace rtow {
namespace {

// Kernel specialisation.  One source, a few instantiations; the host picks by what the scene contains so
// that a scene never pays registers or code for features it does not use:
//   WORLD      0 = BvhNode world (threaded, nodes staged in LDS), 1 = HittableList world (uniform scan),
//              2 = HittableList of static spheres only (config C2: scalar-fed discriminant scan + LDS queue)
//   COMPOSITE  instances / boxes / lists / media may appear as leaves
//   RICH       Perlin-noise or image textures may appear
//   BATCH      BVH world with composite leaves: the leaf phase runs one kind of leaf at a time (deep trees, see walk_leaf_pass)
//   NESTED     REF_TREE leaves may appear: composites kept as the reference's object tree (tree_hit)
//   BLOCK      threads per workgroup.  256 (four waves) everywhere except the deep general kernel, which runs ONE workgroup of
//              768 threads per CU -- three waves per SIMD as before, but one copy of the scene tables in the CU's 160 KB of LDS
//              instead of three in 52 KB each: besides the node rows, the box rows and the sphere table fit as well
//   FAST       primitive-only BVH world walked through the library's own SAH tree, near child first (flat_scene.h FastNodeRec)
//              (with COMPOSITE and BATCH: the segmented walk of a composite world, Traits::SEG)
//   GROUPED    list scan with the leaves of every ray dealt to several lanes (scan_leaves_grouped): launches with pixels_per_wave < 64
#ifndef RT_PARK_STATE
#define RT_PARK_STATE 1
#endif
#ifndef RT_BIG_BLOCK
#define RT_BIG_BLOCK 768  // workgroups this large run one per CU with the scene tables in (nearly) all of its LDS
#endif
template <int WORLD_, bool COMPOSITE_, bool RICH_, int MIN_WAVES_ = 1, bool MEDIA_ = COMPOSITE_, bool BATCH_ = false, bool NESTED_ = false,
          int BLOCK_ = 256, bool FAST_ = false, bool GROUPED_ = false>
struct Traits {
    static constexpr bool GROUPED = GROUPED_ && WORLD_ == 1 && !MEDIA_ && !NESTED_;
    // list scan over instances / boxes compiled for five waves per SIMD (C4, TListInstances5): the path state the scan does not touch
    // -- pixel sum, throughput, emitted light, RNG, pixel counters -- waits in LDS while the leaves are tested (park_* in
    // render_kernel), so that 96 registers are nearly enough (at four waves the same parking costs 2.5 %: measured, not used)
    static constexpr bool PARK = RT_PARK_STATE && WORLD_ == 1 && COMPOSITE_ && !RICH_ && !MEDIA_ && !NESTED_ && !GROUPED_ && MIN_WAVES_ >= 5;
    // segmented walk (flat_scene.h FastOrder / SegMedium): the library's tree over the surface leaves of a composite BVH world,
    // walked once per run of leaves between two media; kind-batched leaf phases, one 768-thread workgroup per CU
    static constexpr bool SEG = FAST_ && COMPOSITE_ && BATCH_ && WORLD_ == 0 && BLOCK_ >= RT_BIG_BLOCK;
    static constexpr bool FAST = FAST_ && !COMPOSITE_ && WORLD_ == 0;
    // kernels that can be launched with heavy / light pixel classes (RenderArgs::heavy_list): for the others the serving code folds away
    static constexpr bool ROLES = WORLD_ == 2 || (FAST_ && WORLD_ == 0 && BLOCK_ >= RT_BIG_BLOCK) || (COMPOSITE_ && BATCH_ && WORLD_ == 0 && BLOCK_ >= RT_BIG_BLOCK);
    static constexpr int BLOCK = BLOCK_;
    static constexpr bool NESTED = NESTED_ && COMPOSITE_;
    static constexpr bool BATCH = BATCH_ && COMPOSITE_ && WORLD_ == 0;
    static constexpr bool MEDIA = MEDIA_;  // ConstantMedium leaves may appear (needs COMPOSITE)
    static constexpr int MIN_WAVES = MIN_WAVES_;  // waves per SIMD the register allocator must leave room for
    static constexpr int WORLD = WORLD_;
    static constexpr bool COMPOSITE = COMPOSITE_;
    static constexpr bool RICH = RICH_;
};

struct Vec {
    double x, y, z;
};
struct Ray {
    Vec o, d;
    double tm;
};
struct HitInfo {
    double t;
    uint32_t ref;  // primitive (or medium) that won
    uint32_t obj;  // enclosing composite object, or kNone
};
struct Surface {
    Vec p, n;
    double u, v;
    uint32_t mat;
    bool front;
};

// BVH nodes as the traversal sees them: 72-byte rows in LDS (or the global 64-byte table when they do not fit).
// The LDS copy is addressed straight off the __shared__ symbol (see lds_node_*), never through generic pointers:
// a pointer that may be LDS or global makes the compiler emit flat loads plus aperture arithmetic per access.
struct NodeView {
    const BvhNodeRec *global;
    uint32_t n;      // number of staged nodes (plane stride)
    bool in_lds;
};

#define DEV __device__ __forceinline__
#define RT_LDS __attribute__((address_space(3)))  // pointers typed for LDS: loads through them are ds_read, never flat

extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

// BVH nodes in LDS: one 72-byte row per node -- six doubles (xlo, xhi, ylo, yhi, zlo, zhi), three words (a, b, escape),
// one word of padding.  One base address per visit and immediate offsets (ds_read2_b64) instead of one address per
// field; the odd multiple of 8 bytes spreads the rows of 64 lanes standing on 64 different nodes over 32 bank
// positions (64-byte rows would fall on 4).
constexpr uint32_t kLdsNodeBytes = 72;
DEV double lds_node_f64(uint32_t, uint32_t field, uint32_t node)
{
    return reinterpret_cast<const double *>(lds_raw + __umul24(node, kLdsNodeBytes))[field];  // node < 2^24: 24-bit multiply
}
DEV uint32_t lds_node_u32(uint32_t, uint32_t word, uint32_t node)
{
    return reinterpret_cast<const uint32_t *>(lds_raw + __umul24(node, kLdsNodeBytes) + 48u)[word];
}

// Scene tables are immutable for the whole launch.  Reading them through the constant address space tells
// the compiler so: a wave-uniform row then always comes through the scalar cache into SGPRs, even though the
// kernel stores pixels inside its main loop (which defeats alias analysis for ordinary global loads).
#define RT_CONST __attribute__((address_space(4)))

// Small tables staged in LDS (DeviceScene::lds_*): a row by byte offset off the same symbol.
template <class R>
DEV R lds_row(uint32_t byte_off, uint32_t idx)
{
    return *reinterpret_cast<const R *>(lds_raw + byte_off + idx * (uint32_t)sizeof(R));
}
// Tables are immutable for the whole launch: read through the constant address space, a wave-uniform index (list
// scans) then comes through the scalar cache, a per-lane one as an ordinary (invariant) vector load.
template <class R>
DEV R const_row(const R *table, uint32_t i)
{
    static_assert(sizeof(R) % 8 == 0, "rows are whole quadwords");
    const RT_CONST uint64_t *src = (const RT_CONST uint64_t *)(uintptr_t)(table + i);
    union {
        R row;
        uint64_t w[sizeof(R) / 8];
    } u;
#pragma unroll
    for (uint32_t k = 0; k < sizeof(R) / 8; k++) u.w[k] = src[k];
    return u.row;
}
// The same through a pointer typed for the LDS address space, for call sites that choose between the LDS copy and the global
// table at run time: two loads through generic pointers get merged into one flat load through a selected pointer.
template <class R>
DEV R lds_row_typed(uint32_t byte_off, uint32_t idx)
{
    static_assert(sizeof(R) % 8 == 0, "rows are whole quadwords");
    const RT_LDS uint64_t *src = (const RT_LDS uint64_t *)(lds_raw + byte_off + idx * (uint32_t)sizeof(R));
    union {
        R row;
        uint64_t w[sizeof(R) / 8];
    } u;
#pragma unroll
    for (uint32_t k = 0; k < sizeof(R) / 8; k++) u.w[k] = src[k];
    return u.row;
}
DEV MSphereGeom get_msphere(const DeviceScene &sc, uint32_t i) { return sc.lds_mspheres != kNone ? lds_row_typed<MSphereGeom>(sc.lds_mspheres, i) : sc.mspheres[i]; }
DEV SphereAux get_msphere_aux(const DeviceScene &sc, uint32_t i) { return sc.lds_msphere_aux != kNone ? lds_row_typed<SphereAux>(sc.lds_msphere_aux, i) : sc.msphere_aux[i]; }
DEV SphereAux get_sphere_aux(const DeviceScene &sc, uint32_t i) { return sc.lds_sphere_aux != kNone ? lds_row_typed<SphereAux>(sc.lds_sphere_aux, i) : sc.sphere_aux[i]; }
DEV SphereGeom get_sphere_typed(const DeviceScene &sc, uint32_t i) { return sc.lds_spheres_tab != kNone ? lds_row_typed<SphereGeom>(sc.lds_spheres_tab, i) : sc.spheres[i]; }
DEV AAQuad get_quad_aa(const DeviceScene &sc, uint32_t i) { return sc.lds_quad_aa != kNone ? lds_row<AAQuad>(sc.lds_quad_aa, i) : const_row(sc.quad_aa, i); }
DEV BoxRec get_box(const DeviceScene &sc, uint32_t i) { return sc.lds_boxes != kNone ? lds_row<BoxRec>(sc.lds_boxes, i) : const_row(sc.boxes, i); }
DEV ObjectRec get_object(const DeviceScene &sc, uint32_t i) { return sc.lds_objects != kNone ? lds_row<ObjectRec>(sc.lds_objects, i) : const_row(sc.objects, i); }
DEV Xform get_xform(const DeviceScene &sc, uint32_t i) { return sc.lds_xforms != kNone ? lds_row<Xform>(sc.lds_xforms, i) : const_row(sc.xforms, i); }
DEV SphereGeom get_sphere(const DeviceScene &sc, uint32_t i) { return sc.lds_spheres_tab != kNone ? lds_row<SphereGeom>(sc.lds_spheres_tab, i) : sc.spheres[i]; }
DEV GroupBox get_group_box(const DeviceScene &sc, uint32_t i) { return sc.lds_group_boxes != kNone ? lds_row<GroupBox>(sc.lds_group_boxes, i) : sc.group_boxes[i]; }
DEV MediumRec get_medium(const DeviceScene &sc, uint32_t i) { return sc.lds_media != kNone ? lds_row<MediumRec>(sc.lds_media, i) : const_row(sc.media, i); }
DEV bool material_needs_uv(const DeviceScene &sc, uint32_t i)
{
    if (sc.lds_materials != kNone)
        return *(const RT_LDS uint32_t *)(lds_raw + sc.lds_materials + i * (uint32_t)sizeof(MaterialRec) + (uint32_t)offsetof(MaterialRec, needs_uv)) != 0;
    return sc.materials[i].needs_uv != 0;
}

template <class T>
DEV const RT_CONST double *const_doubles(const T *p)
{
    return (const RT_CONST double *)(uintptr_t)p;
}
DEV SphereGeom load_sphere_row(const SphereGeom *table, uint32_t k)
{
    const RT_CONST double *p = const_doubles(table + k);
    return SphereGeom{p[0], p[1], p[2], p[3]};
}

DEV Vec mk(double x, double y, double z) { return Vec{x, y, z}; }
DEV Vec operator+(Vec a, Vec b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV Vec operator-(Vec a, Vec b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV Vec operator-(Vec a) { return mk(-a.x, -a.y, -a.z); }
DEV Vec operator*(Vec a, Vec b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV Vec operator*(double t, Vec a) { return mk(t * a.x, t * a.y, t * a.z); }
DEV double dot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV Vec cross(Vec u, Vec v) { return mk(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
DEV double length_sq(Vec a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
DEV double length(Vec a) { return sqrt(length_sq(a)); }
DEV Vec over(Vec a, double t) { return (1 / t) * a; }  // R/Vec3.h:103-106: v / t is (1/t) * v
DEV Vec unit(Vec a) { return over(a, length(a)); }
DEV Vec at(const Ray &r, double t) { return r.o + t * r.d; }
DEV Vec reflect(Vec v, Vec n) { return v - (2.0 * dot(v, n)) * n; }  // R/Vec3.h:127-130
DEV Vec refract(Vec uv, Vec n, double eta)                            // R/Vec3.h:132-141
{
    double ct = fmin(dot(-uv, n), 1.0);
    Vec perp = eta * (uv + ct * n);
    Vec par = (-sqrt(fabs(1.0 - length_sq(perp)))) * n;
    return perp + par;
}

// ------------------------------------------------------------------------------------------------
// primitive tests.  Each returns the accepted t exactly as the reference's Hit would set rec.T.
// ------------------------------------------------------------------------------------------------
// Root selection of R/Sphere.h:36-60 given b, disc (> 0) and a.  The second root is only evaluated when
// the first is at or below tmin: if the first root is >= tmax so is the second (a > 0, s >= 0, and
// division by a is monotone), so the reference's second test cannot pass there either.
DEV bool sphere_roots(double b, double disc, double a, double tmin, double tmax, double &t_out)
{
    double s = sqrt(disc);
    double t = (-b - s) / a;
    if (t < tmax) {
        if (t > tmin) {
            t_out = t;
            return true;
        }
        t = (-b + s) / a;
        if (t < tmax && t > tmin) {
            t_out = t;
            return true;
        }
    }
    return false;
}

// R/Sphere.h:28-60 (also MovingSphere.h:54-86): strict interval, disc > 0, first root then second.
DEV bool sphere_test(Vec oc, Vec d, double a, double r2, double tmin, double tmax, double &t_out)
{
    double b = dot(oc, d);
    double c = dot(oc, oc) - r2;
    double disc = b * b - a * c;
    if (disc > 0.0) {
        // Both roots are <= 0 when the origin is outside (c > 0) and the sphere lies behind (b > 0):
        // -b - s < 0, and s = sqrt(b*b - a*c) <= |b| in fp64 because a*c > 0, so -b + s <= 0.  With
        // tmin >= 0 neither can pass `temp > tmin`; skipping the sqrt/divides changes nothing.
        if (tmin >= 0.0 && b > 0.0 && c > 0.0) return false;
        return sphere_roots(b, disc, a, tmin, tmax, t_out);
    }
    return false;
}

// R/Quad.h:52-99: inclusive interval, inclusive unit square.
DEV bool quad_test(const QuadGeom &q, const Ray &r, double tmin, double tmax, double &t_out)
{
    Vec n = mk(q.nx, q.ny, q.nz);
    double denom = dot(n, r.d);
    if (fabs(denom) < 1e-8) return false;
    double t = (q.d - dot(n, r.o)) / denom;
    if (t < tmin || t > tmax) return false;
    Vec ph = at(r, t) - mk(q.qx, q.qy, q.qz);
    Vec w = mk(q.wx, q.wy, q.wz);
    double alpha = dot(w, cross(ph, mk(q.vx, q.vy, q.vz)));
    double beta = dot(w, cross(mk(q.ux, q.uy, q.uz), ph));
    if (!(0.0 <= alpha && alpha <= 1.0) || !(0.0 <= beta && beta <= 1.0)) return false;
    t_out = t;
    return true;
}

// Axis-aligned quads (AAQuad, flat_scene.h): the same t, alpha and beta as quad_test, without the terms that are exact zeros.
template <int K>
DEV double comp(const Vec &v)
{
    return K == 0 ? v.x : (K == 1 ? v.y : v.z);
}
template <int A>
DEV bool aa_plane(double na, double d, const Ray &r, double tmin, double tmax, double &t)
{
    double denom = na * comp<A>(r.d);
    t = (d - na * comp<A>(r.o)) / denom;
    return !(fabs(denom) < 1e-8) && !(t < tmin || t > tmax);
}
template <int P, int Q>
DEV bool aa_inside(double wa, double qp, double qq, double ku, double kv, const Ray &r, double t)
{
    double php = (comp<P>(r.o) + t * comp<P>(r.d)) - qp;
    double phq = (comp<Q>(r.o) + t * comp<Q>(r.d)) - qq;
    double alpha = wa * (php * kv);
    double beta = wa * (ku * phq);
    return (0.0 <= alpha && alpha <= 1.0) && (0.0 <= beta && beta <= 1.0);
}
template <int A, int P>
DEV bool aa_quad_test(const AAQuad &q, const Ray &r, double tmin, double tmax, double &t_out)
{
    double t;
    if (!aa_plane<A>(q.na, q.d, r, tmin, tmax, t)) return false;
    if (!aa_inside<P, 3 - A - P>(q.wa, q.qp, q.qq, q.ku, q.kv, r, t)) return false;
    t_out = t;
    return true;
}
DEV bool quad_test_at(const DeviceScene &sc, uint32_t idx, const Ray &r, double tmin, double tmax, double &t)
{
    const AAQuad q = get_quad_aa(sc, idx);
    switch (q.code) {
    case 1 + 3 * 0 + 1: return aa_quad_test<0, 1>(q, r, tmin, tmax, t);
    case 1 + 3 * 0 + 2: return aa_quad_test<0, 2>(q, r, tmin, tmax, t);
    case 1 + 3 * 1 + 0: return aa_quad_test<1, 0>(q, r, tmin, tmax, t);
    case 1 + 3 * 1 + 2: return aa_quad_test<1, 2>(q, r, tmin, tmax, t);
    case 1 + 3 * 2 + 0: return aa_quad_test<2, 0>(q, r, tmin, tmax, t);
    case 1 + 3 * 2 + 1: return aa_quad_test<2, 1>(q, r, tmin, tmax, t);
    default: return quad_test(const_row(sc.quads, idx), r, tmin, tmax, t);
    }
}

// MakeBox (R/Instance.h:166-184) as HittableList::Hit sees it (R/HittableList.h:39-57): six faces in list order
// against one running closest-so-far.  The six plane distances do not depend on each other, so they are evaluated
// together (six divides in flight instead of one after another); the interior tests then run in list order.
//
// Interior test of a face with in-plane axes p, q.  R/Quad.h:86-96 accepts when alpha and beta are in [0, 1], where
// (AAQuad) alpha = fl(wa * fl(fl(P[p] - Q[p]) * kv)) = (P[p] - Q[p]) / u[p] * (1 + e), |e| < 2^-49, P the hit point
// as computed.  The host has checked (box_of_six) that Q[p] is mn[p] with u[p] = fl(mx[p] - mn[p]), or mx[p] with the
// negated extent.  With m = 2^-30 (|mn[p]| + |mx[p]|):
//   mn[p] + m <= P[p] <= mx[p] - m  =>  2^-30 <= (P[p] - Q[p]) / u[p] <= 1 - 2^-31  =>  alpha in [0, 1] for certain;
//   P[p] < mn[p] - m or P[p] > mx[p] + m  =>  alpha < 0 or alpha > 1 for certain;
// the same for beta along q.  Only a hit point inside the 2m sliver around an edge needs alpha / beta themselves.
DEV bool box_closest(const DeviceScene &sc, const BoxRec &bx, const Ray &r, double tmin, double tmax, double &t_best,
                     uint32_t &ref_best)
{
    double t[6];
    bool ok[6];
    ok[0] = aa_plane<2>(bx.na[0], bx.d[0], r, tmin, tmax, t[0]);  // front
    ok[1] = aa_plane<0>(bx.na[1], bx.d[1], r, tmin, tmax, t[1]);  // right
    ok[2] = aa_plane<2>(bx.na[2], bx.d[2], r, tmin, tmax, t[2]);  // back
    ok[3] = aa_plane<0>(bx.na[3], bx.d[3], r, tmin, tmax, t[3]);  // left
    ok[4] = aa_plane<1>(bx.na[4], bx.d[4], r, tmin, tmax, t[4]);  // top
    ok[5] = aa_plane<1>(bx.na[5], bx.d[5], r, tmin, tmax, t[5]);  // bottom
    const Vec mn = mk(bx.mn[0], bx.mn[1], bx.mn[2]), mx = mk(bx.mx[0], bx.mx[1], bx.mx[2]);
    const double k30 = 9.313225746154785e-10;  // 2^-30
    const Vec m = mk(k30 * (fabs(mn.x) + fabs(mx.x)), k30 * (fabs(mn.y) + fabs(mx.y)), k30 * (fabs(mn.z) + fabs(mx.z)));
    // (the four bounds from a host-made table through scalar loads instead: measured 1-3 % slower, four and five waves per SIMD)
    const Vec in_lo = mn + m, in_hi = mx - m, out_lo = mn - m, out_hi = mx + m;
    const uint32_t quad_first = bx.quad_first;
    double closest = tmax;
    bool any = false;
#define RT_BOX_FACE(k, P, Q)                                                                                            \
    {                                                                                                                   \
        /* straight-line: every lane evaluates every face (selects); only the sliver case branches */                   \
        const double pp = comp<P>(r.o) + t[k] * comp<P>(r.d), pq = comp<Q>(r.o) + t[k] * comp<Q>(r.d);                  \
        bool accept = pp >= comp<P>(in_lo) && pp <= comp<P>(in_hi) && pq >= comp<Q>(in_lo) && pq <= comp<Q>(in_hi);     \
        const bool outside = pp < comp<P>(out_lo) || pp > comp<P>(out_hi) || pq < comp<Q>(out_lo) || pq > comp<Q>(out_hi); \
        const bool live = ok[k] && !(t[k] > closest);                                                                   \
        if (live && !accept && !outside) {                                                                              \
            const AAQuad f = get_quad_aa(sc, quad_first + k);                                                           \
            accept = aa_inside<P, Q>(f.wa, f.qp, f.qq, f.ku, f.kv, r, t[k]);                                            \
        }                                                                                                               \
        const bool take = live && accept;                                                                               \
        closest = take ? t[k] : closest;                                                                                \
        ref_best = take ? make_ref(REF_QUAD, quad_first + k) : ref_best;                                                \
        any = any || take;                                                                                              \
    }
    RT_BOX_FACE(0, 0, 1)
    RT_BOX_FACE(1, 2, 1)
    RT_BOX_FACE(2, 0, 1)
    RT_BOX_FACE(3, 2, 1)
    RT_BOX_FACE(4, 0, 2)
    RT_BOX_FACE(5, 0, 2)
#undef RT_BOX_FACE
    t_best = closest;
    return any;
}

// unit_time: every row has time0 = 0 and time1 - time0 = 1, so (tm - 0) / 1 == tm exactly and the divide is skipped
DEV Vec msphere_center(const MSphereGeom &g, double tm, bool unit_time = false)  // R/MovingSphere.h:51-52
{
    double frac = unit_time ? tm : (tm - g.t0) / g.dt;
    return mk(g.c0x, g.c0y, g.c0z) + frac * mk(g.dcx, g.dcy, g.dcz);
}

DEV bool prim_test(const DeviceScene &sc, uint32_t ref, const Ray &r, double a, double tmin, double tmax, double &t)
{
    uint32_t idx = ref & kRefIndexMask;
    switch (ref >> kRefShift) {
    case REF_SPHERE: {
        SphereGeom g = sc.lds_spheres_tab != kNone ? lds_row<SphereGeom>(sc.lds_spheres_tab, idx) : const_row(sc.spheres, idx);
        return sphere_test(r.o - mk(g.cx, g.cy, g.cz), r.d, a, g.r2, tmin, tmax, t);
    }
    case REF_MSPHERE: {
        MSphereGeom g = const_row(sc.mspheres, idx);
        return sphere_test(r.o - msphere_center(g, r.tm, (sc.flags & SCENE_MS_UNIT_TIME) != 0), r.d, a, g.r2, tmin, tmax, t);
    }
    default: {
        return quad_test_at(sc, idx, r, tmin, tmax, t);
    }
    }
}

#ifndef RT_PHASES
#define RT_PHASES 0  // diagnostic build: per-phase wave cycles and lane occupancy, summed into ray_counter[8..]
#endif
#if RT_PHASES
#define PH_BEGIN() const unsigned long long ph_t0 = __builtin_readcyclecounter()
#define PH_END(k, cond)                                                   \
    do {                                                                  \
        ph.t[k] += __builtin_readcyclecounter() - ph_t0;                  \
        ph.l[k] += (unsigned long long)__popcll(__ballot(cond));          \
        ph.n[k] += 1ull;                                                  \
    } while (0)
struct PhaseSums {
    // 0 node, 1 leaf, 2 shade, 3 refill; per-lane shares inside divergent code (scaled x1024): 4 group/instance, 5 medium,
    // 6 primitive, 8 record+xforms, 9 box, 10 sub-BVH, 11 other geometry; wave level again: 16 box pass, 17 medium pass,
    // 18 object pass, 19 primitive pass (kind-batched kernels), 20 hit record, 21 scatter, 22 next camera ray, 23 pixel done
    unsigned long long t[24], l[24], n[24];
};
#define PH_ARG , PhaseSums &ph
#define PH_PASS , ph
#define PH_SUB_BEGIN() const unsigned long long ph_s0 = __builtin_readcyclecounter()
// per-lane event counts (summed over the wave at the end like the per-lane time shares): slot 14 = node visits
#define PH_COUNT(k) do { ph.l[k] += 1024ull; ph.n[k] += 1024ull; } while (0)
// inside divergent control flow the sums are per lane: each lane books its share (x1024), the wave adds them up at the end
#define PH_SUB_END(k)                                                                              \
    do {                                                                                           \
        const unsigned long long ph_pc = (unsigned long long)__popcll(__ballot(true));             \
        ph.t[k] += (__builtin_readcyclecounter() - ph_s0) * 1024ull / ph_pc;                       \
        ph.l[k] += 1024ull;                                                                        \
        ph.n[k] += 1024ull / ph_pc;                                                                \
    } while (0)
#else
#define PH_BEGIN() do { } while (0)
#define PH_END(k, cond) do { } while (0)
#define PH_ARG
#define PH_PASS
#define PH_SUB_BEGIN() do { } while (0)
#define PH_COUNT(k) do { } while (0)
#define PH_SUB_END(k) do { } while (0)
#endif
// Ray into the object space of a composite leaf: Translate (R/Instance.h:46) and RotateY (:121-131)
// applied outermost first.
DEV Ray to_object_space(const DeviceScene &sc, const ObjectRec &o, const Ray &r)
{
    Ray lr = r;
    for (uint32_t k = 0; k < o.xf_count; k++) {
        Xform x = get_xform(sc, o.xf_first + k);
        if (x.kind == XF_TRANSLATE) {
            lr.o = lr.o - mk(x.a, x.b, x.c);
        } else {
            double st = x.a, ct = x.b;
            lr.o = mk((ct * lr.o.x) - (st * lr.o.z), lr.o.y, (st * lr.o.x) + (ct * lr.o.z));
            lr.d = mk((ct * lr.d.x) - (st * lr.d.z), lr.d.y, (st * lr.d.x) + (ct * lr.d.z));
        }
    }
    return lr;
}

DEV bool box_test(double xlo, double xhi, double ylo, double yhi, double zlo, double zhi, const Ray &r, Vec inv,
                  double tmin, double tmax);

// Sub-BVH over a large group's primitives (threaded like the world BVH; nodes stay in global memory / L2).
DEV bool subbvh_closest(const DeviceScene &sc, uint32_t root, const Ray &lr, double a, double tmin, double tmax,
                        double &t_best, uint32_t &ref_best)
{
    Vec inv = mk(1.0 / lr.d.x, 1.0 / lr.d.y, 1.0 / lr.d.z);
    double closest = tmax;
    bool any = false;
    uint32_t n = root;
    while (n != kNone) {
        BvhNodeRec node = sc.nodes[n];
        uint32_t next = node.escape;
        if (box_test(node.xlo, node.xhi, node.ylo, node.yhi, node.zlo, node.zhi, lr, inv, tmin, closest)) {
            if ((node.a >> kRefShift) == REF_INNER) {
                next = n + 1;
            } else {
                double t;
                if (prim_test(sc, node.a, lr, a, tmin, closest, t)) {
                    any = true;
                    closest = t;
                    ref_best = node.a;
                }
                if (node.b != node.a && prim_test(sc, node.b, lr, a, tmin, closest, t)) {
                    any = true;
                    closest = t;
                    ref_best = node.b;
                }
            }
        }
        n = next;
    }
    t_best = closest;
    return any;
}

// Closest hit over a composite leaf's geometry (R/HittableList.h:39-57 for groups).
DEV bool geom_closest(const DeviceScene &sc, const ObjectRec &o, const Ray &lr, double tmin, double tmax,
                      double &t_best, uint32_t &ref_best)
{
    double a = dot(lr.d, lr.d);
    bool any = false;
    double closest = tmax;
    if (o.geom_kind == GEOM_BVH) return subbvh_closest(sc, o.first, lr, a, tmin, tmax, t_best, ref_best);
    switch (o.geom_kind) {
    case GEOM_SINGLE: {
        double t;
        if (prim_test(sc, o.first, lr, a, tmin, closest, t)) {
            any = true;
            closest = t;
            ref_best = o.first;
        }
        break;
    }
    case GEOM_SPHERES:
        for (uint32_t k = 0; k < o.count; k++) {
            SphereGeom g = get_sphere(sc, o.first + k);
            double t;
            if (sphere_test(lr.o - mk(g.cx, g.cy, g.cz), lr.d, a, g.r2, tmin, closest, t)) {
                any = true;
                closest = t;
                ref_best = make_ref(REF_SPHERE, o.first + k);
            }
        }
        break;
    case GEOM_MSPHERES:
        for (uint32_t k = 0; k < o.count; k++) {
            MSphereGeom g = sc.mspheres[o.first + k];
            double t;
            if (sphere_test(lr.o - msphere_center(g, lr.tm), lr.d, a, g.r2, tmin, closest, t)) {
                any = true;
                closest = t;
                ref_best = make_ref(REF_MSPHERE, o.first + k);
            }
        }
        break;
    case GEOM_BOX:
        any = box_closest(sc, get_box(sc, o.first), lr, tmin, tmax, closest, ref_best);
        break;
    case GEOM_QUADS:
        for (uint32_t k = 0; k < o.count; k++) {
            double t;
            if (quad_test_at(sc, o.first + k, lr, tmin, closest, t)) {
                any = true;
                closest = t;
                ref_best = make_ref(REF_QUAD, o.first + k);
            }
        }
        break;
    default:
        for (uint32_t k = 0; k < o.count; k++) {
            uint32_t ref = sc.items[o.first + k];
            double t;
            if (prim_test(sc, ref, lr, a, tmin, closest, t)) {
                any = true;
                closest = t;
                ref_best = ref;
            }
        }
        break;
    }
    t_best = closest;
    return any;
}

// Composite leaf: instance chain and, for media, the stochastic volume hit (R/ConstantMedium.h:52-94).
// MED: what the caller knows about the leaf -- 1 a ConstantMedium, 0 not one, -1 look at the object record.
// object_span: the geometric part -- the instance chain, then one closest-hit query over [tmin, tmax] for a surface (t1, pref)
// or, for a medium, the two boundary queries of R/ConstantMedium.h:58-64 (t1, t2, unclipped).  false: nothing hit.
template <class T, int MED>
DEV bool object_span(const DeviceScene &sc, uint32_t oi, const Ray &r, double tmin, double tmax, bool &medium, MediumRec &med, uint32_t &medium_index,
                     double &t1, double &t2, uint32_t &pref PH_ARG)
{
#if RT_PHASES
    const unsigned long long ph_o0 = __builtin_readcyclecounter();
#endif
    ObjectRec o = get_object(sc, oi);
    Ray lr = to_object_space(sc, o, r);
#if RT_PHASES
    {
        // force the loads to land before the stamp
        asm volatile("" ::"v"(lr.o.x), "v"(lr.d.z), "v"(o.first));
        const unsigned long long ph_s0 = ph_o0;
        PH_SUB_END(8);
    }
#endif
    // surfaces: one closest-hit query over [tmin, tmax]; media: two boundary queries (R/ConstantMedium.h:58-64)
    medium = T::MEDIA && (MED < 0 ? o.medium != kNone : MED == 1);
    medium_index = o.medium;
    t1 = 0.0;
    t2 = 0.0;
    pref = kNone;
    if (medium) med = get_medium(sc, o.medium);
    if (medium && o.geom_kind == GEOM_SINGLE && (o.first >> kRefShift) == REF_SPHERE) {
        // The usual boundary: one sphere.  Both queries of R/ConstantMedium.h:58-64 share oc, b, c and the
        // discriminant (R/Sphere.h:28-34); only the root selection (:36-60) runs twice.  A sphere without transforms
        // has its row repeated in the medium record (one table less to chase through).
        SphereGeom g;
        if (med.sphere != kNone) g = SphereGeom{med.cx, med.cy, med.cz, med.r2};
        else g = get_sphere(sc, o.first & kRefIndexMask);
        const Vec oc = lr.o - mk(g.cx, g.cy, g.cz);
        const double a = dot(lr.d, lr.d);
        const double b = dot(oc, lr.d);
        const double c = dot(oc, oc) - g.r2;
        const double disc = b * b - a * c;
        if (!(disc > 0.0)) return false;
        if (!sphere_roots(b, disc, a, -DBL_MAX, DBL_MAX, t1)) return false;
        if (!sphere_roots(b, disc, a, t1 + 0.0001, DBL_MAX, t2)) return false;
    } else {
        for (int pass = 0; pass < 2; pass++) {  // one inlined copy of the geometry query
            const double lo = medium ? (pass == 0 ? -DBL_MAX : t1 + 0.0001) : tmin;
            const double hi = medium ? DBL_MAX : tmax;
            double t;
            PH_SUB_BEGIN();
            const bool got = geom_closest(sc, o, lr, lo, hi, t, pref);
#if RT_PHASES
            asm volatile("" ::"v"(t));
#endif
            PH_SUB_END(o.geom_kind == GEOM_BOX ? 9 : (o.geom_kind == GEOM_BVH ? 10 : 11));
            if (!got) return false;
            if (pass == 0) t1 = t;
            else t2 = t;
            if (!medium) break;
        }
    }
    return true;
}

// The rest of ConstantMedium::Hit (R/ConstantMedium.h:66-93) given its two boundary hits: clip to [tmin, tmax], draw, compare.
DEV bool medium_draw(double neg_inv_density, uint32_t medium_index, uint32_t oi, const Ray &r, double tmin, double tmax, double t1, double t2,
                     HitInfo &best, Xorwow &rng)
{
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return false;
    if (t1 < 0.0) t1 = 0.0;
    double ray_len = length(r.d);
    double inside = (t2 - t1) * ray_len;
    // log(curand_uniform(..)) has a float argument: the float overload is selected on the reference's
    // toolchain; evaluated here as the correctly rounded fp32 log.
    float lg = (float)log((double)xorwow_uniform(rng));
    double hit_dist = neg_inv_density * (double)lg;
    if (hit_dist > inside) return false;
    best.t = t1 + hit_dist / ray_len;
    best.ref = make_ref(REF_MEDIUM, medium_index);
    best.obj = oi;
    return true;
}

template <class T, int MED = -1>
DEV bool object_test(const DeviceScene &sc, uint32_t oi, const Ray &r, double tmin, double tmax, HitInfo &best, Xorwow &rng PH_ARG)
{
    bool medium;
    MediumRec med{};
    uint32_t medium_index, pref;
    double t1, t2;
    if (!object_span<T, MED>(sc, oi, r, tmin, tmax, medium, med, medium_index, t1, t2, pref PH_PASS)) return false;
    if (!medium) {
        best.t = t1;
        best.ref = pref;
        best.obj = oi;
        return true;
    }
    return medium_draw(med.neg_inv_density, medium_index, oi, r, tmin, tmax, t1, t2, best, rng);
}

// ------------------------------------------------------------------------------------------------
// General nesting: a composite kept as the reference's own object tree (flat_scene.h TreeNodeRec), evaluated by making
// the reference's calls in the reference's order with an explicit stack -- Translate::Hit / RotateY::Hit
// (R/Instance.h:41-56,116-150), HittableList::Hit (R/HittableList.h:39-57), ConstantMedium::Hit (R/ConstantMedium.h:52-94:
// two boundary calls, one draw) and BvhNode::Hit (R/BvhNode.h:101-158, incl. BvhNode objects among the leaves, which its
// loop walks as inner nodes).  The hit record is the kernel's usual (t, primitive, where): the point, normal and their
// way back through the transforms are built once per bounce by make_surface from the winning node's chain.  A failed
// call never touches the record (as in the reference, where every Hit writes only on success); a medium, whose boundary
// calls write a record of their own, keeps the caller's in its frame.
// ------------------------------------------------------------------------------------------------
struct TreeFrame {
    uint32_t node;
    uint32_t step;     // LIST: next child; MEDIUM / transforms: 0, 1, 2; BVH: 0 visit, 1 after leaf a, 2 after leaf b
    uint32_t cursor;   // BVH: threaded node
    uint32_t any;
    double tmin, tmax; // the interval this call was made with
    double closest;    // LIST / BVH: closestSoFar;  MEDIUM: t of the first boundary hit
    HitInfo saved;     // MEDIUM: the caller's record
};

DEV Ray chain_ray(const DeviceScene &sc, uint32_t first, uint32_t count, const Ray &r)
{
    Ray lr = r;
    for (uint32_t k = 0; k < count; k++) {
        Xform x = get_xform(sc, first + k);
        if (x.kind == XF_TRANSLATE) {
            lr.o = lr.o - mk(x.a, x.b, x.c);
        } else {
            double st = x.a, ct = x.b;
            lr.o = mk((ct * lr.o.x) - (st * lr.o.z), lr.o.y, (st * lr.o.x) + (ct * lr.o.z));
            lr.d = mk((ct * lr.d.x) - (st * lr.d.z), lr.d.y, (st * lr.d.x) + (ct * lr.d.z));
        }
    }
    return lr;
}

template <class T>
DEV bool tree_hit(const DeviceScene &sc, uint32_t root, const Ray &r, double tmin0, double tmax0, HitInfo &best, Xorwow &rng)
{
    TreeFrame stack[kTreeMaxDepth];
    int sp = 0;
    stack[0] = TreeFrame{root, 0u, kNone, 0u, tmin0, tmax0, tmax0, HitInfo{0.0, kNone, kNone}};
    bool ret = false;            // what the call that has just returned answered
    uint32_t chain_first = kNone, chain_count = 0;  // the chain `lr` was computed for
    Ray lr = r;
    for (;;) {
        TreeFrame &f = stack[sp];
        const TreeNodeRec n = sc.tree_nodes[f.node];
        if (n.chain_first != chain_first || n.chain_count != chain_count) {  // always from the world ray: same bits every time
            lr = chain_ray(sc, n.chain_first, n.chain_count, r);
            chain_first = n.chain_first;
            chain_count = n.chain_count;
        }
        uint32_t call = kNone;   // child to call next, with [call_tmin, call_tmax]
        double call_tmin = 0.0, call_tmax = 0.0;
        bool answer = false;  // what this call answers, unless it first calls a child (`call`)
        switch (n.kind) {
        case TN_PRIM: {
            double t;
            answer = prim_test(sc, n.a, lr, dot(lr.d, lr.d), f.tmin, f.tmax, t);
            if (answer) {
                best.t = t;
                best.ref = n.a;
                best.obj = kTreeObjBit | f.node;
            }
            break;
        }
        case TN_TRANSLATE:
        case TN_ROTATE_Y:
            if (f.step == 0) {
                f.step = 1;
                call = n.a; call_tmin = f.tmin; call_tmax = f.tmax;
            } else {
                answer = ret;
            }
            break;
        case TN_LIST:
            if (f.step > 0 && ret) {
                f.any = 1;
                f.closest = best.t;
            }
            if (f.step < n.b) {
                call = sc.tree_items[n.a + f.step];
                call_tmin = f.tmin; call_tmax = f.closest;
                f.step++;
            } else {
                answer = f.any != 0;
            }
            break;
        case TN_MEDIUM: {
            if (f.step == 0) {
                f.saved = best;
                f.step = 1;
                call = n.a; call_tmin = -DBL_MAX; call_tmax = DBL_MAX;
            } else if (f.step == 1) {
                if (!ret) {
                    best = f.saved;
                } else {
                    f.closest = best.t;  // rec1.T
                    f.step = 2;
                    call = n.a; call_tmin = f.closest + 0.0001; call_tmax = DBL_MAX;
                }
            } else {
                double t1 = f.closest, t2 = best.t;
                const bool second = ret;
                best = f.saved;
                if (second) {
                    if (t1 < f.tmin) t1 = f.tmin;
                    if (t2 > f.tmax) t2 = f.tmax;
                    if (t1 < t2) {
                        if (t1 < 0.0) t1 = 0.0;
                        const MediumRec med = get_medium(sc, n.b);
                        const double ray_len = length(lr.d);
                        const double inside = (t2 - t1) * ray_len;
                        const float lg = (float)log((double)xorwow_uniform(rng));  // see object_test
                        const double hit_dist = med.neg_inv_density * (double)lg;
                        if (!(hit_dist > inside)) {
                            best.t = t1 + hit_dist / ray_len;
                            best.ref = make_ref(REF_MEDIUM, n.b);
                            best.obj = kTreeObjBit | f.node;
                            answer = true;
                        }
                    }
                }
            }
            break;
        }
        default: {  // TN_BVH
            if (f.cursor == kNone) f.cursor = n.a;  // first entry: the root of this BvhNode's threaded nodes
            for (;;) {
                const BvhNodeRec node = sc.tree_bvh[f.cursor];
                if (f.step == 0) {
                    const Vec inv = mk(1.0 / lr.d.x, 1.0 / lr.d.y, 1.0 / lr.d.z);
                    if (!box_test(node.xlo, node.xhi, node.ylo, node.yhi, node.zlo, node.zhi, lr, inv, f.tmin, f.closest)) {
                        if (node.escape == kNone) {
                            answer = f.any != 0;
                            break;
                        }
                        f.cursor = node.escape;
                        continue;
                    }
                    f.step = 1;
                    if ((node.a >> kRefShift) != REF_INNER) {
                        call = node.a & kRefIndexMask; call_tmin = f.tmin; call_tmax = f.closest;
                        break;
                    }
                    ret = false;
                }
                if (f.step == 1) {
                    if ((node.a >> kRefShift) != REF_INNER && ret) {
                        f.any = 1;
                        f.closest = best.t;
                    }
                    f.step = 2;
                    if ((node.b >> kRefShift) != REF_INNER) {
                        call = node.b & kRefIndexMask; call_tmin = f.tmin; call_tmax = f.closest;
                        break;
                    }
                    ret = false;
                }
                // step 2: both children have been dealt with
                if ((node.b >> kRefShift) != REF_INNER && ret) {
                    f.any = 1;
                    f.closest = best.t;
                }
                f.step = 0;
                const bool has_inner = (node.a >> kRefShift) == REF_INNER || (node.b >> kRefShift) == REF_INNER;
                const uint32_t next = has_inner ? f.cursor + 1u : node.escape;
                if (next == kNone) {
                    answer = f.any != 0;
                    break;
                }
                f.cursor = next;
            }
            break;
        }
        }
        if (call != kNone) {
            if (sp + 1 >= (int)kTreeMaxDepth) return false;  // cannot happen: rt_scene_commit refuses deeper trees
            sp++;
            stack[sp] = TreeFrame{call, 0u, kNone, 0u, call_tmin, call_tmax, call_tmax, HitInfo{0.0, kNone, kNone}};
            ret = false;
            continue;
        }
        // done: return `answer` to the caller
        ret = answer;
        if (sp == 0) return ret;
        sp--;
    }
}

// a leaf whose test may draw random numbers (tagged by the flattener): a ConstantMedium object, or a tree (may hold media)
DEV bool is_medium_leaf(uint32_t ref) { return (ref >> kRefShift) == REF_MOBJECT || (ref >> kRefShift) == REF_TREE; }

template <class T>
DEV bool leaf_test(const DeviceScene &sc, uint32_t ref, const Ray &r, double a, double tmin, double tmax, HitInfo &best, Xorwow &rng PH_ARG)
{
    if constexpr (T::COMPOSITE) {
        if ((ref >> kRefShift) == REF_BOX) {
            PH_SUB_BEGIN();
            double t;
            uint32_t face = kNone;
            const bool found = box_closest(sc, get_box(sc, ref & kRefIndexMask), r, tmin, tmax, t, face);
            if (found) {
                best.t = t;
                best.ref = face;
                best.obj = kNone;
            }
            PH_SUB_END(9);
            return found;
        }
        if ((ref >> kRefShift) == REF_OBJECT || (ref >> kRefShift) == REF_MOBJECT) {
            PH_SUB_BEGIN();
            const bool found = object_test<T>(sc, ref & kRefIndexMask, r, tmin, tmax, best, rng PH_PASS);
            PH_SUB_END(is_medium_leaf(ref) ? 5 : 4);
            return found;
        }
    }
    if constexpr (T::NESTED) {
        if ((ref >> kRefShift) == REF_TREE) return tree_hit<T>(sc, ref & kRefIndexMask, r, tmin, tmax, best, rng);
    }
    PH_SUB_BEGIN();
    double t;
    const bool found = prim_test(sc, ref, r, a, tmin, tmax, t);
    if (found) {
        best.t = t;
        best.ref = ref;
        best.obj = kNone;
    }
    PH_SUB_END(6);
    return found;
}


// ------------------------------------------------------------------------------------------------
// world traversal
// ------------------------------------------------------------------------------------------------
// Slab test, R/AABB.h:68-98, with 1/d hoisted out of the node loop (same value every time).
DEV bool box_test(double xlo, double xhi, double ylo, double yhi, double zlo, double zhi, const Ray &r, Vec inv,
                  double tmin, double tmax)
{
    double t0 = (xlo - r.o.x) * inv.x, t1 = (xhi - r.o.x) * inv.x;
    tmin = fmax(tmin, fmin(t0, t1));
    tmax = fmin(tmax, fmax(t0, t1));
    t0 = (ylo - r.o.y) * inv.y;
    t1 = (yhi - r.o.y) * inv.y;
    tmin = fmax(tmin, fmin(t0, t1));
    tmax = fmin(tmax, fmax(t0, t1));
    t0 = (zlo - r.o.z) * inv.z;
    t1 = (zhi - r.o.z) * inv.z;
    tmin = fmax(tmin, fmin(t0, t1));
    tmax = fmin(tmax, fmax(t0, t1));
    return tmax > tmin;
}

// Stackless walk in the reference's visiting order (R/BvhNode.h:101-158): a node's leaf children are
// tested where the node is visited; "pop" is the escape link.  The walk is resumable: a lane keeps its
// position (node index, closest hit so far) in registers, so that the kernel can interleave traversal
// bursts with shading and never makes 63 lanes wait for the one ray that visits 150 nodes.
constexpr uint32_t kWalkParked = 0x80000000u;
DEV bool walk_moving(uint32_t state) { return (int32_t)state >= 0; }
DEV bool walk_parked(uint32_t state) { return (int32_t)state < -1; }

struct Walk {
    Vec inv;          // 1 / direction (R/AABB.h:77,84,91 recompute it per node; same value)
    double a;         // dot(d, d)
    double closest;
    // Where the lane stands, in one word so that each of the wave's three questions is a single compare:
    //   moving  (state >= 0 as int32): `state` is the next node to visit;
    //   parked  (bit 31 set, not kNone): standing on bottom node `state & 0x7FFFFFFF` whose box was hit -- its leaves
    //           are tested in the next leaf phase;
    //   done    (kNone): the walk is complete (or no walk is in progress).
    uint32_t state;
    uint32_t oct_off;  // library-tree kernels: byte offset, inside a FastNodeF row, of the link pair of this ray's direction octant
    // library-tree kernels: the ray as the fp32 slab test of their node boxes wants it (box_test_f): 1 / direction and the
    // origin, in fp32 (`inv` above is then unused: those boxes are never tested in fp64)
    float ivx, ivy, ivz, cx, cy, cz;  // 1 / direction; origin
    bool any;
};

template <bool LIBRARY_TREE = false>
DEV void walk_begin(Walk &w, const Ray &r, double tmax)
{
    w.oct_off = 32u + 4u * ((r.d.x < 0.0 ? 1u : 0u) | (r.d.y < 0.0 ? 2u : 0u) | (r.d.z < 0.0 ? 4u : 0u));
    if constexpr (LIBRARY_TREE) {
        w.ivx = 1.0f / (float)r.d.x; w.ivy = 1.0f / (float)r.d.y; w.ivz = 1.0f / (float)r.d.z;
        w.cx = (float)r.o.x; w.cy = (float)r.o.y; w.cz = (float)r.o.z;
    } else {
        w.inv = mk(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    }
    w.a = dot(r.d, r.d);
    w.closest = tmax;
    w.state = 0;
    w.any = false;
}

DEV uint32_t leaf_kind(uint32_t ref);
constexpr uint32_t kWalkKindShift = 28;  // kind-batched kernels: bits 28-29 of a parked state = kind of the pending leaf

// One inner-node step: box test, then descend / escape; a bottom node whose box is hit parks the lane.
template <bool BATCH = false>
DEV void walk_node(const NodeView &nv, const Ray &r, double tmin, Walk &w)
{
    const uint32_t n = w.state;
    double xlo, xhi, ylo, yhi, zlo, zhi;
    uint32_t na, next;
    if (nv.in_lds) {
        xlo = lds_node_f64(nv.n, 0, n); xhi = lds_node_f64(nv.n, 1, n); ylo = lds_node_f64(nv.n, 2, n);
        yhi = lds_node_f64(nv.n, 3, n); zlo = lds_node_f64(nv.n, 4, n); zhi = lds_node_f64(nv.n, 5, n);
        na = lds_node_u32(nv.n, 0, n); next = lds_node_u32(nv.n, 2, n);
    } else {
        const BvhNodeRec *node = nv.global + n;
        xlo = node->xlo; xhi = node->xhi; ylo = node->ylo; yhi = node->yhi; zlo = node->zlo; zhi = node->zhi;
        na = node->a; next = node->escape;
    }
    // selects, not branches: hit an inner node -> its first child; hit a bottom node -> park on it until the leaf phase;
    // missed -> the escape link
    const bool hit = box_test(xlo, xhi, ylo, yhi, zlo, zhi, r, w.inv, tmin, w.closest);
    uint32_t park = n | kWalkParked;
    if constexpr (BATCH) park |= leaf_kind(na) << kWalkKindShift;  // the wave sorts its parked lanes by this, without a table read
    const uint32_t down = (na >> kRefShift) == REF_INNER ? n + 1u : park;
    w.state = hit ? down : next;
}

// Leaf phase for a parked lane: the bottom node's one or two leaves, in the reference's order.
template <class T>
DEV void walk_leaves(const DeviceScene &sc, const NodeView &nv, const Ray &r, double tmin, Walk &w, HitInfo &best, Xorwow &rng PH_ARG)
{
    const uint32_t n = w.state & ~kWalkParked;
    uint32_t na, nb, next;
    if (nv.in_lds) {
        na = lds_node_u32(nv.n, 0, n); nb = lds_node_u32(nv.n, 1, n); next = lds_node_u32(nv.n, 2, n);
    } else {
        const BvhNodeRec *node = nv.global + n;
        na = node->a; nb = node->b; next = node->escape;
    }
    // span-1 nodes hold the same leaf twice (R/BvhNode.h:63-67).  Re-testing a surface with
    // tmax = its own t changes nothing; a medium draws again, so only media are re-tested.
    bool again = nb != na;
    if constexpr (T::MEDIA) again = again || is_medium_leaf(nb);
    if constexpr (!T::COMPOSITE) {
        // The common bottom node of a sphere world: two moving-sphere rows (static spheres are stored as such rows too,
        // see unify_spheres).  Both rows are fetched together -- one round trip to L2 instead of two -- and tested in
        // the reference's order.
        if ((na >> kRefShift) == REF_MSPHERE && (nb >> kRefShift) == REF_MSPHERE) {
            const bool unit_time = (sc.flags & SCENE_MS_UNIT_TIME) != 0;
            const MSphereGeom ga = sc.mspheres[na & kRefIndexMask], gb = sc.mspheres[nb & kRefIndexMask];
            double t;
            if (sphere_test(r.o - msphere_center(ga, r.tm, unit_time), r.d, w.a, ga.r2, tmin, w.closest, t)) {
                w.any = true;
                w.closest = t;
                best.t = t;
                best.ref = na;
                best.obj = kNone;
            }
            if (again && sphere_test(r.o - msphere_center(gb, r.tm, unit_time), r.d, w.a, gb.r2, tmin, w.closest, t)) {
                w.any = true;
                w.closest = t;
                best.t = t;
                best.ref = nb;
                best.obj = kNone;
            }
            w.state = next;
            return;
        }
    }
    for (int c = 0; c < 2; c++) {  // one inlined copy of the leaf test
        if (c == 1 && !again) break;
        if (leaf_test<T>(sc, c ? nb : na, r, w.a, tmin, w.closest, best, rng PH_PASS)) {
            w.any = true;
            w.closest = best.t;
        }
    }
    w.state = next;
}

// ---- the library's own tree (FastNodeF): same resumable walk, 68-byte rows, links chosen by the ray's octant ----
constexpr uint32_t kFastNodeBytes = (uint32_t)sizeof(FastNodeF);
static_assert(sizeof(FastNodeF) == 68 && offsetof(FastNodeF, a) == 24 && offsetof(FastNodeF, link) == 32, "row layout: box 24, leaf refs 8, links 32, pad 4");
// Conservative slab test in fp32.  The boxes of the library's tree only prune: a ray that passes one it should not merely
// visits a node in vain, a ray that FAILS one it should pass would lose a hit.  So the test may err only towards "hit":
//   t = (lo - o) * (1/d) like R/AABB.h:77-97, in fp32.  Its error against the exact (lo - o) / d is at most
//   2^-22 (|lo| + |o|) / |d| (the fp32 copies of o and of 1/d, the subtraction, the product), and the rows are 2^-19 of the
//   scene's reach (>= |lo|, |o|) wider than the boxes (device_scene.cpp): several times that; the ray's interval
//   [tmin, closest] is a little wider too.
// A zero direction component gives infinities, and a NaN where the origin lies in the plane: v_min / v_max drop NaNs, an
// infinite bound never cuts the interval short on the wrong side -- the behaviour of R/AABB.h:68-98 in fp64.  (The cheaper
// lo * (1/d) - o * (1/d), one fma per plane, is NOT safe: with 1/d infinite both terms are, and inf - inf or -inf - inf
// discards or inverts the slab -- measured: 1.6 % of C5's pixels at 5000 spp, from directions with a component exactly 0.)
DEV bool box_test_f(float xlo, float xhi, float ylo, float yhi, float zlo, float zhi, const Walk &w, float tmin, double closest)
{
    const float t0x = (xlo - w.cx) * w.ivx, t1x = (xhi - w.cx) * w.ivx;
    const float t0y = (ylo - w.cy) * w.ivy, t1y = (yhi - w.cy) * w.ivy;
    const float t0z = (zlo - w.cz) * w.ivz, t1z = (zhi - w.cz) * w.ivz;
    const float cf = (float)closest * 1.000002f;  // >= closest (round to nearest loses at most 2^-24); +inf stays +inf
    const float tnear = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
    const float tfar = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), cf));
    return tfar > tnear;
}
DEV bool fast_row_hit(uint32_t base, const Walk &w, double closest)
{
    const RT_LDS float *b = (const RT_LDS float *)(lds_raw + base);
    return box_test_f(b[0], b[1], b[2], b[3], b[4], b[5], w, 0.000999f, closest);  // tmin = 0.001, a little early
}
// The rows are only ever read from LDS (the launcher falls back to the reference-tree kernel when they do not fit): a
// branch between an LDS and a global copy made the compiler merge the two into generic pointers and flat loads.
[[maybe_unused]] constexpr uint32_t kFastLdsBudget = 60 * 1024;
DEV void walk_node_fast(const Ray &r, double tmin, Walk &w)
{
    const uint32_t n = w.state;
    const uint32_t base = __umul24(n, kFastNodeBytes);
    const uint32_t links = *reinterpret_cast<const uint32_t *>(lds_raw + base + w.oct_off);
    const bool hit = fast_row_hit(base, w, w.closest);
    const uint32_t first = links & 0xFFFFu, esc = links >> 16;
    const uint32_t down = (first & kFastBottom) ? (n | kWalkParked) : first;  // a bottom node's hit link says "park here"
    const uint32_t next = esc == kFastEnd ? kNone : esc;
    w.state = hit ? down : next;
}
// The same for the segmented walk's usual case -- no limit on the leaf positions: the parked state carries the kind of the
// pending leaf (kind-batched leaf phases), which the bottom node's hit link holds in the place the state wants it.
static_assert(kFastBottom << 16 == kWalkParked && kWalkKindShift == 28, "hit link of a bottom node << 16 = parked bit | kind");
DEV void walk_node_open(const Ray &r, double tmin, Walk &w)
{
    const uint32_t n = w.state;
    const uint32_t base = __umul24(n, kFastNodeBytes);
    const uint32_t links = *reinterpret_cast<const uint32_t *>(lds_raw + base + w.oct_off);
    const bool hit = fast_row_hit(base, w, w.closest);
    const uint32_t first = links & 0xFFFFu, esc = links >> 16;
    const uint32_t down = (first & kFastBottom) ? (n | (first << 16)) : first;
    const uint32_t next = esc == kFastEnd ? kNone : esc;
    w.state = hit ? down : next;
}
DEV void walk_leaves_fast(const DeviceScene &sc, const Ray &r, double tmin, Walk &w, HitInfo &best)
{
    const uint32_t n = w.state & ~kWalkParked;
    const uint32_t base = __umul24(n, kFastNodeBytes);
    const uint32_t na = *reinterpret_cast<const uint32_t *>(lds_raw + base + 24u);
    const uint32_t nb = *reinterpret_cast<const uint32_t *>(lds_raw + base + 28u);
    const uint32_t links = *reinterpret_cast<const uint32_t *>(lds_raw + base + w.oct_off);
    const uint32_t esc = links >> 16;
    const uint32_t next = esc == kFastEnd ? kNone : esc;
    if ((na >> kRefShift) == REF_MSPHERE && nb != kNone && (nb >> kRefShift) == REF_MSPHERE) {  // both rows in one round trip
        const bool unit_time = (sc.flags & SCENE_MS_UNIT_TIME) != 0;
        const MSphereGeom ga = get_msphere(sc, na & kRefIndexMask), gb = get_msphere(sc, nb & kRefIndexMask);
        double t;
        if (sphere_test(r.o - msphere_center(ga, r.tm, unit_time), r.d, w.a, ga.r2, tmin, w.closest, t)) {
            w.any = true; w.closest = t; best.t = t; best.ref = na; best.obj = kNone;
        }
        if (sphere_test(r.o - msphere_center(gb, r.tm, unit_time), r.d, w.a, gb.r2, tmin, w.closest, t)) {
            w.any = true; w.closest = t; best.t = t; best.ref = nb; best.obj = kNone;
        }
        w.state = next;
        return;
    }
    if ((na >> kRefShift) == REF_MSPHERE && nb == kNone) {  // a bottom node of one sphere row
        const MSphereGeom ga = get_msphere(sc, na & kRefIndexMask);
        double t;
        if (sphere_test(r.o - msphere_center(ga, r.tm, (sc.flags & SCENE_MS_UNIT_TIME) != 0), r.d, w.a, ga.r2, tmin, w.closest, t)) {
            w.any = true; w.closest = t; best.t = t; best.ref = na; best.obj = kNone;
        }
        w.state = next;
        return;
    }
    for (int c = 0; c < 2; c++) {  // one inlined copy of the test
        const uint32_t ref = c ? nb : na;
        if (ref == kNone) break;
        double t;
        if (prim_test(sc, ref, r, w.a, tmin, w.closest, t)) {
            w.any = true; w.closest = t; best.t = t; best.ref = ref; best.obj = kNone;
        }
    }
    w.state = next;
}

// ---- segmented walk of a composite world (Traits::SEG) ----------------------------------------------------------------
// The same near-child-first walk over 88-byte rows, restricted to the leaves whose position in the reference's visiting
// order is below `hi` (the walk that leads up to a medium: what follows the medium in that order must not be seen yet):
// a node all of whose leaves come later counts as a box miss.  A bottom node parks the lane on its first leaf that is
// allowed (kWalkSecond set when that is leaf b).  Positions have no lower limit: a leaf met again by a later walk of the
// same ray answers as before or not at all (its test is repeated with the closest hit it helped to find).
DEV uint32_t leaf_kind(uint32_t ref);
constexpr uint32_t kWalkSecond = 0x40000000u;  // parked on the bottom node's SECOND leaf
DEV void walk_node_seg(const DeviceScene &sc, const Ray &r, double tmin, Walk &w, uint32_t hi)
{
    const uint32_t n = w.state;
    const uint32_t base = __umul24(n, kFastNodeBytes);
    const uint32_t na = *reinterpret_cast<const uint32_t *>(lds_raw + base + 24u);
    const uint32_t nb = *reinterpret_cast<const uint32_t *>(lds_raw + base + 28u);
    const uint32_t links = *reinterpret_cast<const uint32_t *>(lds_raw + base + w.oct_off);
    const uint32_t span = *reinterpret_cast<const uint32_t *>(lds_raw + sc.lds_fast_order + n * 8u);        // omin | omax << 16
    const uint32_t leaves = *reinterpret_cast<const uint32_t *>(lds_raw + sc.lds_fast_order + n * 8u + 4u);  // oa | ob << 16
    const bool inner = (na >> kRefShift) == REF_INNER;
    const uint32_t oa = leaves & 0xFFFFu, ob = leaves >> 16;
    const bool a_in = oa < hi, b_in = nb != kNone && ob < hi;
    const bool in_range = inner ? (span & 0xFFFFu) < hi : (a_in || b_in);
    const bool hit = in_range && fast_row_hit(base, w, w.closest);
    const uint32_t first = links & 0xFFFFu, esc = links >> 16;
    const uint32_t pending = a_in ? na : nb;
    const uint32_t park = n | kWalkParked | (a_in ? 0u : kWalkSecond) | (leaf_kind(pending) << kWalkKindShift);
    const uint32_t next = esc == kFastEnd ? kNone : esc;
    w.state = hit ? (inner ? first : park) : next;
}
// the pending leaf of a lane parked by walk_node_seg
DEV uint32_t seg_pending_leaf(uint32_t state)
{
    return *reinterpret_cast<const uint32_t *>(lds_raw + __umul24(state & 0x0FFFFFFFu, kFastNodeBytes) + ((state & kWalkSecond) ? 28u : 24u));
}
struct SegState {
    uint32_t lo, hi;  // the walk in progress (or just completed) covers the leaf positions [lo, hi); hi == kSegEnd: the ray's last walk
    uint32_t stage;   // next medium (index into seg_media) the ray has to deal with
};

// Composite worlds: the leaf phase runs one KIND of leaf at a time.  A box, a medium, an instance and a plain primitive
// are four different pieces of code; tested in one divergent pass the wave executes each of them with the few lanes
// that happen to stand on that kind (measured on the Book-2 final scene: 12-16 of 64 lanes per pass).  So a parked
// lane remembers which of its bottom node's two leaves is pending, the wave counts the pending leaves by kind, and a
// kind is run when enough lanes wait for it (or the walkers have run out, or this is the last round before the next
// look at the shading queue).  Each lane still meets its own leaves in the reference's order -- only when the wave
// executes them changes -- so the RNG draws of media are consumed exactly as before.
enum : uint32_t { LK_BOX = 0u, LK_MEDIUM = 1u, LK_OBJECT = 2u, LK_PRIM = 3u };
constexpr uint32_t kWalkNodeMask = 0x0FFFFFFFu;
DEV uint32_t leaf_kind(uint32_t ref)
{
    const uint32_t tag = ref >> kRefShift;
    return tag == REF_BOX ? LK_BOX : (tag == REF_MOBJECT ? LK_MEDIUM : (tag == REF_OBJECT ? LK_OBJECT : LK_PRIM));
}
// the pending leaf of a parked lane
DEV uint32_t walk_pending_leaf(const NodeView &nv, uint32_t state)
{
    const uint32_t n = state & kWalkNodeMask, word = (state & kWalkSecond) ? 1u : 0u;
    if (nv.in_lds) return lds_node_u32(nv.n, word, n);
    return word ? nv.global[n].b : nv.global[n].a;
}

template <class T, uint32_t K>
DEV bool leaf_test_kind(const DeviceScene &sc, uint32_t ref, const Ray &r, double a, double tmin, double tmax, HitInfo &best, Xorwow &rng PH_ARG)
{
    if constexpr (K == LK_BOX) {
        double t;
        uint32_t face = kNone;
        const bool found = box_closest(sc, get_box(sc, ref & kRefIndexMask), r, tmin, tmax, t, face);
        if (found) {
            best.t = t;
            best.ref = face;
            best.obj = kNone;
        }
        return found;
    } else if constexpr (K == LK_MEDIUM) {
        return object_test<T, 1>(sc, ref & kRefIndexMask, r, tmin, tmax, best, rng PH_PASS);
    } else if constexpr (K == LK_OBJECT) {
        return object_test<T, 0>(sc, ref & kRefIndexMask, r, tmin, tmax, best, rng PH_PASS);
    } else {
        double t;
        const bool found = prim_test(sc, ref, r, a, tmin, tmax, t);
        if (found) {
            best.t = t;
            best.ref = ref;
            best.obj = kNone;
        }
        return found;
    }
}

// One pass over the lanes whose pending leaf `ref` is of kind K: test it; if the node's other leaf is due (span-1 nodes
// hold the same leaf twice, R/BvhNode.h:63-67: only a medium is tested again, see walk_leaves) and is of the same kind
// it is tested in the same pass, otherwise the lane stays parked on it until that kind's pass.
template <class T, uint32_t K>
DEV void walk_leaf_pass(const DeviceScene &sc, const NodeView &nv, const Ray &r, double tmin, Walk &w, HitInfo &best, Xorwow &rng,
                        uint32_t ref PH_ARG, bool have_first = false, bool first_found = false, double first_t = 0.0,
                        uint32_t first_ref = kNone, uint32_t seg_lo = 0u, uint32_t seg_hi = 0u)
{
    const uint32_t n = w.state & kWalkNodeMask;
    uint32_t nb, next;
    [[maybe_unused]] bool b_in = false;  // segmented walk: the node's second leaf lies in the interval being walked
    if constexpr (T::SEG) {
        const uint32_t base = __umul24(n, kFastNodeBytes);
        nb = *reinterpret_cast<const uint32_t *>(lds_raw + base + 28u);
        const uint32_t esc = *reinterpret_cast<const uint32_t *>(lds_raw + base + w.oct_off) >> 16;
        next = esc == kFastEnd ? kNone : esc;
        const uint32_t ob = *reinterpret_cast<const uint32_t *>(lds_raw + sc.lds_fast_order + n * 8u + 4u) >> 16;
        b_in = nb != kNone && ob < seg_hi;
    } else if (nv.in_lds) {
        nb = lds_node_u32(nv.n, 1, n); next = lds_node_u32(nv.n, 2, n);
    } else {
        nb = nv.global[n].b; next = nv.global[n].escape;
    }
    bool second = (w.state & kWalkSecond) != 0;
    for (int c = 0; c < 2; c++) {  // one inlined copy of the test
        bool found;
        if (c == 0 && have_first) {  // the wave has already answered this one together (walk_object_pass)
            found = first_found;
            if (found) {
                best.t = first_t;
                best.ref = first_ref;
                best.obj = ref & kRefIndexMask;
            }
        } else {
            found = leaf_test_kind<T, K>(sc, ref, r, w.a, tmin, w.closest, best, rng PH_PASS);
        }
        if (found) {
            w.any = true;
            w.closest = best.t;
        }
        bool again = !second && nb != ref;
        if constexpr (T::SEG) again = !second && b_in;  // no leaf is held twice in the library's tree, and none is a medium
        else if constexpr (T::MEDIA) again = again || (!second && is_medium_leaf(nb));
        if (!again) {
            w.state = next;
            break;
        }
        if (leaf_kind(nb) != K) {
            w.state = (w.state & ~(3u << kWalkKindShift)) | kWalkSecond | (leaf_kind(nb) << kWalkKindShift);
            break;
        }
        second = true;
        ref = nb;
    }
}

// The LK_OBJECT pass.  A large group of static spheres behind an instance (the Book-2 final scene's 1000-sphere
// cluster) is not walked lane by lane through its sub-BVH in global memory -- a handful of lanes chasing ~50 dependent
// L2 loads each while the other sixty wait -- but scanned by the whole wave, one parked ray at a time: lane l tests
// spheres l, l + 64, ... of the group in object space and a wave-wide minimum over (t, index) picks the hit.  Spheres
// draw no random numbers and the reference's list scan keeps the smallest acceptable root (lowest index on ties), so the
// result is the one HittableList::Hit returns (R/HittableList.h:39-57; the same argument as scan_cooperative's).
// Everything else of this kind (small groups, boxes behind transforms, ...) takes the per-lane test.
DEV double bcast(double x, int src_lane);
DEV void wave_min(double &t, uint32_t &k);
template <class T>
DEV void walk_object_pass(const DeviceScene &sc, const NodeView &nv, const Ray &r, double tmin, Walk &w, HitInfo &best, Xorwow &rng,
                          uint32_t pending, bool mine, uint32_t lane PH_ARG, uint32_t seg_lo = 0u, uint32_t seg_hi = 0u)
{
    ObjectRec o{};
    Ray lr = r;
    o.coop_first = kNone;
    if (mine) {
        o = get_object(sc, pending & kRefIndexMask);
        if (o.coop_first != kNone) lr = to_object_space(sc, o, r);
    }
    const bool coop = mine && o.coop_first != kNone;
    bool found = false;
    double found_t = 0.0;
    uint32_t found_ref = kNone;
    unsigned long long todo = __ballot(coop);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        Ray q;
        q.o = mk(bcast(lr.o.x, src), bcast(lr.o.y, src), bcast(lr.o.z, src));
        q.d = mk(bcast(lr.d.x, src), bcast(lr.d.y, src), bcast(lr.d.z, src));
        const double tmax = bcast(w.closest, src);
        const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)o.coop_first, src);
        const uint32_t count = (uint32_t)__builtin_amdgcn_readlane((int)o.count, src);
        const uint32_t boxes = (uint32_t)__builtin_amdgcn_readlane((int)o.coop_boxes, src);
        const double a = dot(q.d, q.d);
        double bt = tmax;
        uint32_t bk = kNone;
        // Cull by groups of sixteen rows: lane g slab-tests the padded box of group g (R/AABB.h's test, conservative here),
        // then the wave tests only the spheres of the groups the ray passes, four groups (64 spheres) at a time.  A zero
        // direction component would put NaNs into the slab test: such a ray (none in practice) culls nothing.
        const bool axis_parallel = q.d.x == 0.0 || q.d.y == 0.0 || q.d.z == 0.0;
        const Vec inv = mk(1.0 / q.d.x, 1.0 / q.d.y, 1.0 / q.d.z);
        const uint32_t n_groups = (count + kCoopGroup - 1u) / kCoopGroup;
        for (uint32_t g0 = 0; g0 < n_groups; g0 += 64u) {
            const uint32_t g = g0 + lane;
            bool pass = g < n_groups;
            if (pass && !axis_parallel) {
                const GroupBox gb = get_group_box(sc, boxes + g);
                pass = box_test(gb.lo[0], gb.hi[0], gb.lo[1], gb.hi[1], gb.lo[2], gb.hi[2], q, inv, tmin, tmax);
            }
            unsigned long long hits = __ballot(pass);
            while (hits) {
                // the (lane / 16)-th of the next four passing groups is mine
                unsigned long long m = hits;
                int mine_g = -1;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int bit = m ? __ffsll((long long)m) - 1 : -1;
                    if ((int)(lane >> 4) == j) mine_g = bit;
                    m = m ? (m & (m - 1)) : 0ull;
                }
                hits = m;
                const uint32_t k = mine_g >= 0 ? (g0 + (uint32_t)mine_g) * kCoopGroup + (lane & 15u) : count;
                if (k < count) {
                    const SphereGeom sg = get_sphere(sc, first + k);
                    const Vec oc = q.o - mk(sg.cx, sg.cy, sg.cz);
                    const double b = dot(oc, q.d);
                    const double c = dot(oc, oc) - sg.r2;
                    const double disc = b * b - a * c;
                    if (disc > 0.0 && !(tmin >= 0.0 && b > 0.0 && c > 0.0)) {  // see sphere_test
                        double t;
                        if (sphere_roots(b, disc, a, tmin, bt, t)) {
                            bt = t;
                            bk = k;
                        }
                    }
                }
            }
        }
        wave_min(bt, bk);
        if ((int)lane == src) {
            found = bk != kNone;
            found_t = bt;
            found_ref = make_ref(REF_SPHERE, first + bk);
        }
    }
    if (mine) walk_leaf_pass<T, LK_OBJECT>(sc, nv, r, tmin, w, best, rng, pending PH_PASS, coop, found, found_t, found_ref, seg_lo, seg_hi);
}

// Segmented walk: what a ray does between two walks.  Called when the lane has no walk in progress -- a new ray, or the walk
// over the leaf positions [sg.lo, sg.hi) has just ended.  It deals with the world's media in visiting order:
//   * a medium whose padded bounding box the ray does not cross between 0 and the closest hit so far cannot answer:
//     ConstantMedium::Hit clips its two boundary hits to [tMin, tMax] and returns false, before its draw, unless t1 < t2
//     is left (R/ConstantMedium.h:66-71) -- and both boundary hits lie inside the box.  The reference's call would return
//     false without a draw, so the medium is skipped and the surfaces on both sides of it are one segment;
//   * the same when its own boundary queries and clipping, evaluated with the closest hit so far, already say so
//     (object_span + the clipping of R/ConstantMedium.h:66-70; the closest hit can only come nearer, which clips more);
//   * otherwise the surfaces that precede it are walked first, as far as they can matter (their closest hit is the tMax the
//     reference calls the medium with), then the medium is tested -- twice where the reference's span-1 node holds it
//     twice -- with the reference's own arithmetic and draws (object_test);
//   * after the last medium, one walk over the rest of the list.
// The reference reaches a medium only through its BVH, i.e. only if the boxes of the nodes above it are hit within the
// closest hit so far; a medium inside a box the ray misses, or one that lies wholly beyond the closest hit, returns false
// before its draw here as well (no boundary hit, or t1 >= t2 after clipping to tMax), so the draws are the same ones.
template <class T>
DEV void seg_advance(const DeviceScene &sc, const Ray &ray, Walk &w, HitInfo &best, Xorwow &rng, SegState &sg PH_ARG)
{
    uint32_t lo = sg.hi, hi, stage = sg.stage;
    w.closest = w.any ? best.t : DBL_MAX;  // a walk that led up to a medium ran under a lowered bound (below): back to the closest hit itself
    for (;;) {
        if (stage >= sc.n_seg_media) {
            hi = kSegEnd;
            break;
        }
        const SegMedium m = lds_row<SegMedium>(sc.lds_seg_media, stage);
        // its boundary queries (no draw yet): would the call get past its clipping with the closest hit so far?
        bool is_medium;
        MediumRec med{};
        uint32_t medium_index, pref;
        double t1, t2;
        if (!box_test_f(m.fbox[0], m.fbox[1], m.fbox[2], m.fbox[3], m.fbox[4], m.fbox[5], w, 0.0f, w.closest) ||
            !object_span<T, 1>(sc, m.object, ray, 0.001, w.closest, is_medium, med, medium_index, t1, t2, pref PH_PASS) ||
            (t1 < 0.001 ? 0.001 : t1) >= (t2 > w.closest ? w.closest : t2)) {
            stage++;  // it would return false before its draw: no medium here for this ray
            continue;
        }
        // A ray whose stretch that still matters lies inside the medium's box (scattered inside the medium: both ends inside, the box
        // is convex) can only hit the surface leaves that reach into the box: the host has listed them (SegMedium candidates).
        auto inside = [&](Vec p) {
            return p.x >= m.lo[0] && p.x <= m.hi[0] && p.y >= m.lo[1] && p.y <= m.hi[1] && p.z >= m.lo[2] && p.z <= m.hi[2];
        };
        const bool local = m.cand_count != kNone && inside(ray.o);
        // One loop, two trips, so that the candidate tests exist once in the code: trip 0 = the candidates that precede the
        // medium (only when the walk up to it can be skipped), then the medium's draws; trip 1 = all candidates, when the ray
        // ends inside the box after the last medium.
        bool walk_first = false, tested_before = false;
#pragma nounroll
        for (int trip = 0; trip < 2; trip++) {
            const bool here = trip == 0 ? (lo < m.order && local && inside(at(ray, fmin(w.closest, t2))))
                                        : (stage >= sc.n_seg_media && local && w.any && inside(at(ray, w.closest)));
            // trip 0: the candidates before the medium; trip 1: the others -- and those of trip 0 again only if that trip did not run
            // (a leaf answers the same t every time: tested again under a closer bound it can only fall away)
            const uint32_t from = (trip == 1 && tested_before) ? m.order : 0u, below = trip == 0 ? m.order : kSegEnd;
            if (here) {
                for (uint32_t c = 0; c < m.cand_count; c++) {
                    const SegCandidate cd = lds_row<SegCandidate>(sc.lds_seg_cand, m.cand_first + c);
                    if (cd.order < from || cd.order >= below) continue;
                    bool found;
                    if ((cd.ref >> kRefShift) == REF_BOX) {
                        // a box: its two corners first, as a slab test a part in 2^30 wider than the box (the six face planes are the
                        // corner coordinates up to their last bits) -- most candidates end here
                        const BoxRec bx = get_box(sc, cd.ref & kRefIndexMask);
                        const double k30 = 9.313225746154785e-10;
                        const Vec pad = mk(k30 * (fabs(bx.mn[0]) + fabs(bx.mx[0])), k30 * (fabs(bx.mn[1]) + fabs(bx.mx[1])), k30 * (fabs(bx.mn[2]) + fabs(bx.mx[2])));
                        found = box_test(bx.mn[0] - pad.x, bx.mx[0] + pad.x, bx.mn[1] - pad.y, bx.mx[1] + pad.y, bx.mn[2] - pad.z, bx.mx[2] + pad.z, ray,
                                         mk(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z), 0.0, w.closest);
                        if (found) {
                            double t;
                            uint32_t face = kNone;
                            found = box_closest(sc, bx, ray, 0.001, w.closest, t, face);
                            if (found) {
                                best.t = t;
                                best.ref = face;
                                best.obj = kNone;
                            }
                        }
                    } else {
                        found = leaf_test_kind<T, LK_PRIM>(sc, cd.ref, ray, w.a, 0.001, w.closest, best, rng PH_PASS);
                    }
                    if (found) {
                        w.any = true;
                        w.closest = best.t;
                    }
                }
                if (trip == 0) tested_before = true;
                else lo = kSegEnd;  // the candidates were all there is left to hit: no walk
            }
            if (trip == 1) break;
            if (lo < m.order && !here) {
                // The surfaces before it first -- but only as far as the medium's far side: the closest hit among them is the tMax
                // of the medium's call, which clips the far boundary hit to it (t2 = min(t2, tMax)); a hit beyond t2 changes
                // nothing there, and the ray's last walk finds it again.
                walk_first = true;
                break;
            }
            for (uint32_t c = 0; c <= m.twice; c++)  // the same two boundary hits every time (geometry), the closest hit as it stands
                if (medium_draw(med.neg_inv_density, medium_index, m.object, ray, 0.001, w.closest, t1, t2, best, rng)) {
                    w.any = true;
                    w.closest = best.t;
                }
            lo = m.order + 1u;
            stage++;
        }
        if (walk_first) {
            hi = m.order;
            w.closest = fmin(w.closest, t2);
            break;
        }
    }
    sg.lo = lo;
    sg.hi = hi;
    sg.stage = stage;
    w.state = (lo < hi && lo < sc.n_world_items) ? 0u : kNone;  // from the root, or nothing to walk
}

// HittableList world (R/HittableList.h:39-57): the item index is wave-uniform, so the primitive rows
// are fetched through the scalar path.
template <class T>
DEV bool world_hit_list(const DeviceScene &sc, const Ray &r, double tmin, double tmax, HitInfo &best, Xorwow &rng PH_ARG)
{
    double a = dot(r.d, r.d);
    double closest = tmax;
    bool any = false;
    const uint32_t n = sc.n_world_items;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t ref = (uint32_t)__builtin_amdgcn_readfirstlane((int)((const RT_CONST uint32_t *)(uintptr_t)sc.world_items)[k]);
        if (leaf_test<T>(sc, ref, r, a, tmin, closest, best, rng PH_PASS)) {
            any = true;
            closest = best.t;
        }
    }
    return any;
}

// HittableList of static spheres (config C2).  Two phases per ray, identical results to the sequential scan:
//  1. convergent scan: every lane evaluates b, c, disc of sphere k (k wave-uniform: the sphere row comes
//     through the scalar cache into SGPRs); a lane for which the reference could accept a root
//     (disc > 0 and not "outside and behind") appends k to its own queue in LDS;
//  2. each lane walks its own queue in ascending k and does the sqrt / divide root selection against its
//     running closest-so-far -- the same order the reference's loop meets those spheres in.
constexpr int kQueueCap = 16;   // entries per lane; the queue is drained whenever a lane could overflow
#ifndef RT_SIMPLE_BREAK
#define RT_SIMPLE_BREAK 0
#endif
#ifndef RT_PROBE_ON
#define RT_PROBE_ON 1
#endif
#ifndef RT_ORDER_ON
#define RT_ORDER_ON 1
#endif
#ifndef RT_STAMP
#define RT_STAMP 0  // diagnostic build: wall-clock stamps of queue exhaustion / first and last wave exit
#endif
#ifndef RT_FAST_VISITS
#define RT_FAST_VISITS 2  // node visits of the library-tree walk per look at the wave's state
#endif
#ifndef RT_BURST
#define RT_BURST 8
#endif
#ifndef RT_ROUNDS
#define RT_ROUNDS 6
#endif
constexpr int kBurst = RT_BURST;    // BVH worlds: at most this many node visits between two leaf phases
constexpr int kRounds = RT_ROUNDS;  // node/leaf phase pairs per look at the shading queue, primitive worlds
constexpr int kRoundsComposite = 4; // the same for composite worlds
constexpr int kRoundsFast = 4;      // and for the library-tree kernel, whose frame ends with its long pixels: they are shaded sooner
                                    // (C3, one call: 2 rounds 2615, 3: 2711, 4: 2891-2943, 5: 2805, 6: 2784, 8: 2675 Msamples/s)

DEV void drain_queue(const SphereGeom *__restrict__ spheres, const uint16_t *queue, uint32_t lane, uint32_t &count,
                     const Ray &r, double a, double tmin, double &closest, uint32_t &best_k)
{
    for (uint32_t s = 0; s < count; s++) {
        uint32_t k = queue[s * 64u + lane];
        SphereGeom g = spheres[k];
        Vec oc = r.o - mk(g.cx, g.cy, g.cz);
        double b = dot(oc, r.d);
        double c = dot(oc, oc) - g.r2;
        double disc = b * b - a * c;
        double t;
        if (sphere_roots(b, disc, a, tmin, closest, t)) {
            closest = t;
            best_k = k;
        }
    }
    count = 0;
}

// Four spheres of the convergent scan: the four discriminant chains are independent straight-line code
// (the scheduler interleaves them), then each lane queues the spheres for which a root could be accepted.
DEV void scan_four(const SphereGeom &g0, const SphereGeom &g1, const SphereGeom &g2, const SphereGeom &g3, uint32_t k0,
                   const Ray &r, double a, uint16_t *queue, uint32_t lane, uint32_t &count)
{
    const SphereGeom *g[4] = {&g0, &g1, &g2, &g3};
    double b[4], c[4], disc[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        Vec oc = r.o - mk(g[u]->cx, g[u]->cy, g[u]->cz);
        b[u] = dot(oc, r.d);
        c[u] = dot(oc, oc) - g[u]->r2;
        disc[u] = b[u] * b[u] - a * c[u];
    }
    // Pin the four chains ahead of the branches: without this the compiler sinks each chain next to its own branch
    // and a wave executes them one after another, every fp64 op waiting for the previous one's result.
    asm volatile("" : "+v"(disc[0]), "+v"(disc[1]), "+v"(disc[2]), "+v"(disc[3]));
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (disc[u] > 0.0 && !(b[u] > 0.0 && c[u] > 0.0)) {  // tmin = 0.001 >= 0: see sphere_test
            queue[count * 64u + lane] = (uint16_t)(k0 + u);
            count++;
        }
    }
}

DEV void scan_one(const SphereGeom &g, uint32_t k, const Ray &r, double a, uint16_t *queue, uint32_t lane, uint32_t &count)
{
    Vec oc = r.o - mk(g.cx, g.cy, g.cz);
    double b = dot(oc, r.d);
    double c = dot(oc, oc) - g.r2;
    double disc = b * b - a * c;
    if (disc > 0.0 && !(b > 0.0 && c > 0.0)) {
        queue[count * 64u + lane] = (uint16_t)k;
        count++;
    }
}

// A no-op that *uses* four scalar rows: it makes the compiler place the wait for those rows here, i.e. before
// the next batch of scalar loads is issued.  (Scalar loads return out of order, so every wait is lgkmcnt(0); a wait
// placed after the next loads were issued would also wait for them and undo the prefetch.)
DEV void rows_arrived(const SphereGeom &g0, const SphereGeom &g1, const SphereGeom &g2, const SphereGeom &g3)
{
    asm volatile("" ::"s"(g0.cx), "s"(g0.cy), "s"(g0.cz), "s"(g0.r2), "s"(g1.cx), "s"(g1.cy), "s"(g1.cz), "s"(g1.r2));
    asm volatile("" ::"s"(g2.cx), "s"(g2.cy), "s"(g2.cz), "s"(g2.r2), "s"(g3.cx), "s"(g3.cy), "s"(g3.cz), "s"(g3.r2));
}

// ---- conservative filter of the list scan -------------------------------------------------------------------------
// The reference tests a ray against a sphere with b = oc.d, c = oc.oc - r^2, disc = b*b - a*c (R/Sphere.h:28-41): 13
// fp64 instructions with the compare, for every sphere of the list, and all but a few per ray end at `disc > 0` false.
// Those are only *rejections*, so they need not be computed the reference's way -- any test that never rejects a sphere
// the reference would accept leaves the image bit for bit the same, as long as the survivors then go through the
// reference's arithmetic (drain_queue).  With u = d/|d|, od = o.u, p = o - od u (the ray's foot point) and, per sphere,
// K = |C|^2 - r^2 (host, scene_builder.cpp):
//     disc / a  =  ((o-C).u)^2 - (|o-C|^2 - r^2)  =  (C.u)^2 + 2 p.C - K  -  (o.o - od^2)
// i.e. Q = fma(s, s, t) with s = C.u (3 instructions), t = 2p.C - (o.o - od^2) (3 fma), compared with the sphere's K: 8
// instructions (the per-ray term is the addend that opens the chain and K is an operand of the compare: either as an
// addend of its own would cost a move from the scalar row into a vector register).
// Rounding: every term above is at most W^2 in magnitude, W = |o| + max(|C| + r) (`scan_reach`); the 8 operations, the
// rounding of u (three divisions and a square root, so u is neither exactly unit nor exactly parallel to d), of K, p and
// od, and the reference's own rounding of disc (divided by a) each move the comparison by at most a few u W^2,
// u = 2^-53; their sum is below 64 u W^2.  The filter lowers the threshold by M = 2^-40 W^2 (8192 u W^2): it passes
// whenever the reference's disc, however rounded, is positive, and a sphere of radius r is passed in vain only by rays
// that miss it by less than M / 2r (4e-6 W^2 / r world units^2: for config C2, W = 2013, a band of 1e-5 of a small
// sphere's radius).  A ray with a non-finite or vanishing direction passes every sphere (u = p = 0, addend +inf).
// The reference's second shortcut -- the sphere is behind an outside origin, sphere_test -- is applied to the survivors
// with the same margin: bu = od - s > sqrt(M) and bu^2 - disc/a > M imply b > 0 and c > 0 in the reference's arithmetic.
struct ScanRay {
    Vec u, p2;           // d / |d|,  2 (o - (o.u) u)
    double nthr;         // M - (o.o - (o.u)^2)
    double od, root_m;   // o.u,  sqrt(M)
};
DEV ScanRay scan_ray(const Ray &r, double a, double reach)
{
    ScanRay f;
    const double oo = dot(r.o, r.o);
    const double w = sqrt(oo) + reach;
    const bool sane = a > 1e-280 && a < 1e280 && w < 1e140;
    if (sane) {
        const double inv = 1.0 / sqrt(a);
        f.u = inv * r.d;
        f.od = dot(r.o, f.u);
        f.p2 = 2.0 * (r.o - f.od * f.u);
        f.root_m = 0x1p-20 * w;
        f.nthr = f.root_m * f.root_m - (oo - f.od * f.od);
    } else {
        f.u = f.p2 = mk(0.0, 0.0, 0.0);
        f.od = 0.0;
        f.root_m = __builtin_inf();
        f.nthr = __builtin_inf();
    }
    return f;
}
DEV double filter_s(const ScanRay &f, double cx, double cy, double cz)
{
    return __builtin_fma(cz, f.u.z, __builtin_fma(cy, f.u.y, cx * f.u.x));
}
DEV double filter_q(const ScanRay &f, double s, double cx, double cy, double cz)  // the sphere passes when this exceeds its K
{
    return __builtin_fma(s, s, __builtin_fma(f.p2.z, cz, __builtin_fma(f.p2.y, cy, __builtin_fma(f.p2.x, cx, f.nthr))));
}
DEV bool filter_behind(const ScanRay &f, double s, double q, double k)  // only for a sphere that passed: q > k
{
    const double bu = f.od - s;
    return bu > f.root_m && __builtin_fma(bu, bu, k - q) > 0.0;
}

DEV SphereScanRow load_scan_row(const SphereScanRow *table, uint32_t k)
{
    const RT_CONST double *p = const_doubles(table + k);
    return SphereScanRow{p[0], p[1], p[2], p[3]};
}
DEV void scan_rows_arrived(const SphereScanRow &g0, const SphereScanRow &g1, const SphereScanRow &g2, const SphereScanRow &g3)
{
    asm volatile("" ::"s"(g0.cx), "s"(g0.cy), "s"(g0.cz), "s"(g0.k), "s"(g1.cx), "s"(g1.cy), "s"(g1.cz), "s"(g1.k));
    asm volatile("" ::"s"(g2.cx), "s"(g2.cy), "s"(g2.cz), "s"(g2.k), "s"(g3.cx), "s"(g3.cy), "s"(g3.cz), "s"(g3.k));
}

// Four spheres through the filter: four independent chains, one branch for the four of them.
DEV void filter_four(const SphereScanRow &g0, const SphereScanRow &g1, const SphereScanRow &g2, const SphereScanRow &g3, uint32_t k0,
                     const ScanRay &f, uint16_t *queue, uint32_t lane, uint32_t &count)
{
    const SphereScanRow *g[4] = {&g0, &g1, &g2, &g3};
    double s[4], q[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        s[u] = filter_s(f, g[u]->cx, g[u]->cy, g[u]->cz);
        q[u] = filter_q(f, s[u], g[u]->cx, g[u]->cy, g[u]->cz);
    }
    asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    const bool p0 = q[0] > g0.k, p1 = q[1] > g1.k, p2 = q[2] > g2.k, p3 = q[3] > g3.k;
    if (p0 | p1 | p2 | p3) {
        const bool p[4] = {p0, p1, p2, p3};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (p[u] && !filter_behind(f, s[u], q[u], g[u]->k)) {
                queue[count * 64u + lane] = (uint16_t)(k0 + u);
                count++;
            }
        }
    }
}

// drain_queue for survivors of the filter: the reference's whole test, `disc > 0` included.  ROWS_IN_LDS: the rows come
// from the LDS planes of the cooperative scan (a queue entry costs one LDS round trip instead of one to L2).
template <bool ROWS_IN_LDS>
DEV void drain_filtered(const SphereGeom *__restrict__ spheres, uint32_t planes_off, uint32_t n_padded, const uint16_t *queue, uint32_t lane,
                        uint32_t &count, const Ray &r, double a, double tmin, double &closest, uint32_t &best_k)
{
    for (uint32_t s = 0; s < count; s++) {
        uint32_t k = queue[s * 64u + lane];
        SphereGeom g;
        if constexpr (ROWS_IN_LDS) {
            const RT_LDS double *pl = (const RT_LDS double *)(lds_raw + planes_off);
            g = SphereGeom{pl[k], pl[n_padded + k], pl[2u * n_padded + k], pl[3u * n_padded + k]};
        } else {
            g = spheres[k];
        }
        double t;
        if (sphere_test(r.o - mk(g.cx, g.cy, g.cz), r.d, a, g.r2, tmin, closest, t)) {
            closest = t;
            best_k = k;
        }
    }
    count = 0;
}

// ---- the same filter in packed fp32, two spheres per instruction (r3) -------------------------------------------------------
// gfx950 issues v_pk_fma_f32 (two fp32 fmas per lane) in the slot of one v_fma_f64, so the filter's seven multiply-adds cost
// 7 instructions per PAIR of spheres plus two compares: 4.5 per sphere instead of 8.  It only ever rejects, so it may be as
// coarse as fp32 makes it as long as it never rejects what the reference accepts.  Error of the computed Q against the exact
// disc / a + K + M (same quantities as above, everything rounded to fp32 -- centre, u, 2p, the per-ray addend, every product
// and sum): at most 2^-24 (17 W^2 + 5 M) with W = |o| + max |C| over the spheres this filter DECIDES (`scan_reach32`: the bulk
// of the list; a sphere far outside it, like the Book-1 ground sphere, has k = -inf and always goes on to the exact test),
// plus 2^-24 W^2 for the rounding of K.  With M = 2^-18 W^2 (64 x 2^-24 W^2) the threshold is lowered 3.5 times that:
// a sphere the reference could accept always passes; a sphere is passed in vain by rays that miss it by less than
// M / 2r (C2: W = 30, M = 3.4e-3, 4 % of a small sphere's radius).  The behind-the-origin shortcut keeps its form with
// the fp32 margins (bu > 2^-9 W >> its own error 2^-21 W; bu^2 - disc/a > M).  tests/test_filter_margin.py restates both
// forms operation by operation.
typedef float v2f __attribute__((ext_vector_type(2)));
struct ScanRay32 {
    float ux, uy, uz, px, py, pz;  // d / |d|,  2 (o - (o.u) u)
    float nthr;                    // M - (o.o - (o.u)^2)
    float od, root_m;              // o.u,  sqrt(M)
};
DEV ScanRay32 scan_ray32(const Ray &r, double a, double reach)
{
    ScanRay32 f;
    const double oo = dot(r.o, r.o);
    const double w = sqrt(oo) + reach;
    const bool sane = a > 1e-280 && a < 1e280 && w < 1e15;
    if (sane) {
        const double inv = 1.0 / sqrt(a);
        const Vec u = inv * r.d;
        const double od = dot(r.o, u);
        const Vec p2 = 2.0 * (r.o - od * u);
        const double root_m = 0x1p-9 * w;
        f.ux = (float)u.x; f.uy = (float)u.y; f.uz = (float)u.z;
        f.px = (float)p2.x; f.py = (float)p2.y; f.pz = (float)p2.z;
        f.od = (float)od;
        f.root_m = (float)root_m;
        f.nthr = (float)(root_m * root_m - (oo - od * od));
    } else {  // a degenerate ray passes every sphere and none is called behind
        f.ux = f.uy = f.uz = f.px = f.py = f.pz = f.od = 0.0f;
        f.root_m = __builtin_inff();
        f.nthr = __builtin_inff();
    }
    return f;
}
DEV SphereScanPair load_scan_pair(const SphereScanPair *table, uint32_t k)
{
    const RT_CONST float *p = (const RT_CONST float *)(uintptr_t)(table + k);
    return SphereScanPair{{p[0], p[1]}, {p[2], p[3]}, {p[4], p[5]}, {p[6], p[7]}};
}
DEV void scan_pairs_arrived(const SphereScanPair &g0, const SphereScanPair &g1)
{
    asm volatile("" ::"s"(g0.cx[0]), "s"(g0.cx[1]), "s"(g0.cy[0]), "s"(g0.cy[1]), "s"(g0.cz[0]), "s"(g0.cz[1]), "s"(g0.k[0]), "s"(g0.k[1]));
    asm volatile("" ::"s"(g1.cx[0]), "s"(g1.cx[1]), "s"(g1.cy[0]), "s"(g1.cy[1]), "s"(g1.cz[0]), "s"(g1.cz[1]), "s"(g1.k[0]), "s"(g1.k[1]));
}
// Two pairs = four spheres (list positions k0 .. k0 + 3): two packed chains, four compares, one branch for the four of them.
DEV void filter_pairs(const SphereScanPair &g0, const SphereScanPair &g1, uint32_t k0, const ScanRay32 &f, uint16_t *queue, uint32_t lane,
                      uint32_t &count)
{
    const v2f ux = {f.ux, f.ux}, uy = {f.uy, f.uy}, uz = {f.uz, f.uz}, px = {f.px, f.px}, py = {f.py, f.py}, pz = {f.pz, f.pz}, nt = {f.nthr, f.nthr};
    const SphereScanPair *g[2] = {&g0, &g1};
    v2f s[2], q[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const v2f cx = {g[h]->cx[0], g[h]->cx[1]}, cy = {g[h]->cy[0], g[h]->cy[1]}, cz = {g[h]->cz[0], g[h]->cz[1]};
        s[h] = __builtin_elementwise_fma(cz, uz, __builtin_elementwise_fma(cy, uy, cx * ux));
        q[h] = __builtin_elementwise_fma(s[h], s[h], __builtin_elementwise_fma(pz, cz, __builtin_elementwise_fma(py, cy, __builtin_elementwise_fma(px, cx, nt))));
    }
    const bool p0 = q[0].x > g0.k[0], p1 = q[0].y > g0.k[1], p2 = q[1].x > g1.k[0], p3 = q[1].y > g1.k[1];
    if (p0 | p1 | p2 | p3) {
        const bool p[4] = {p0, p1, p2, p3};
        const float sv[4] = {s[0].x, s[0].y, s[1].x, s[1].y}, qv[4] = {q[0].x, q[0].y, q[1].x, q[1].y};
        const float kv[4] = {g0.k[0], g0.k[1], g1.k[0], g1.k[1]};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float bu = f.od - sv[u];
            const bool behind = bu > f.root_m && __builtin_fmaf(bu, bu, kv[u] - qv[u]) > 0.0f;  // k = -inf (always passes): never behind
            if (p[u] && !behind) {
                queue[count * 64u + lane] = (uint16_t)(k0 + u);
                count++;
            }
        }
    }
}

// Pixel-parallel scan through the packed fp32 filter: pairs of sphere rows are wave-uniform (scalar path), four pairs (eight
// spheres) per trip in two register sets like scan_filtered; the survivors go through drain_filtered, i.e. the reference's test.
template <bool ROWS_IN_LDS>
DEV bool scan_filtered32(const DeviceScene &sc, uint32_t planes_off, uint32_t n_padded, uint16_t *queue, uint32_t lane, const Ray &r, double tmin,
                         double tmax, HitInfo &best)
{
    const SphereScanPair *__restrict__ rows = sc.sphere_scan32;
    const SphereGeom *__restrict__ spheres = sc.spheres;
    const uint32_t n = sc.n_spheres;
    const uint32_t n_pairs = (n + 1u) >> 1, n4 = n_pairs & ~3u;  // the last pair of an odd list is padded with a row that never passes
    const double a = dot(r.d, r.d);
    const ScanRay32 f = scan_ray32(r, a, sc.scan_reach32);
    double closest = tmax;
    uint32_t best_k = kNone, count = 0;
    SphereScanPair a0{}, a1{};
    if (n4) {
        a0 = load_scan_pair(rows, 0);
        a1 = load_scan_pair(rows, 1);
    }
    for (uint32_t k0 = 0; k0 < n4; k0 += 4) {
        scan_pairs_arrived(a0, a1);
        const SphereScanPair b0 = load_scan_pair(rows, k0 + 2), b1 = load_scan_pair(rows, k0 + 3);
        filter_pairs(a0, a1, 2u * k0, f, queue, lane, count);
        const uint32_t kn = (k0 + 4 < n4) ? k0 + 4 : k0;  // last trip re-reads its own rows (stays in bounds)
        scan_pairs_arrived(b0, b1);
        a0 = load_scan_pair(rows, kn);
        a1 = load_scan_pair(rows, kn + 1);
        filter_pairs(b0, b1, 2u * k0 + 4u, f, queue, lane, count);
        if (__any(count > (uint32_t)(kQueueCap - 8))) drain_filtered<ROWS_IN_LDS>(spheres, planes_off, n_padded, queue, lane, count, r, a, tmin, closest, best_k);
    }
    for (uint32_t k = n4; k < n_pairs; k++) {  // up to three pairs left: one at a time, the second half of the call idle
        const SphereScanPair g = load_scan_pair(rows, k);
        const float inf = __builtin_inff();
        const SphereScanPair none{{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {inf, inf}};
        filter_pairs(g, none, 2u * k, f, queue, lane, count);
        if (__any(count > (uint32_t)(kQueueCap - 4))) drain_filtered<ROWS_IN_LDS>(spheres, planes_off, n_padded, queue, lane, count, r, a, tmin, closest, best_k);
    }
    drain_filtered<ROWS_IN_LDS>(spheres, planes_off, n_padded, queue, lane, count, r, a, tmin, closest, best_k);
    if (best_k == kNone) return false;
    best.t = closest;
    best.ref = make_ref(REF_SPHERE, best_k);
    best.obj = kNone;
    return true;
}

// Pixel-parallel scan through the filter: sphere rows are wave-uniform (scalar path), eight per trip in two register sets
// as in scan_uniform below.
template <bool ROWS_IN_LDS>
DEV bool scan_filtered(const DeviceScene &sc, uint32_t planes_off, uint32_t n_padded, uint16_t *queue, uint32_t lane, const Ray &r, double tmin,
                       double tmax, HitInfo &best)
{
    const SphereScanRow *__restrict__ rows = sc.sphere_scan;
    const SphereGeom *__restrict__ spheres = sc.spheres;
    const uint32_t n = sc.n_spheres;
    const uint32_t n8 = n & ~7u;
    const double a = dot(r.d, r.d);
    const ScanRay f = scan_ray(r, a, sc.scan_reach);
    double closest = tmax;
    uint32_t best_k = kNone, count = 0;
    SphereScanRow a0{}, a1{}, a2{}, a3{};
    if (n8) {
        a0 = load_scan_row(rows, 0); a1 = load_scan_row(rows, 1);
        a2 = load_scan_row(rows, 2); a3 = load_scan_row(rows, 3);
    }
    for (uint32_t k0 = 0; k0 < n8; k0 += 8) {
        scan_rows_arrived(a0, a1, a2, a3);
        const SphereScanRow b0 = load_scan_row(rows, k0 + 4), b1 = load_scan_row(rows, k0 + 5);
        const SphereScanRow b2 = load_scan_row(rows, k0 + 6), b3 = load_scan_row(rows, k0 + 7);
        filter_four(a0, a1, a2, a3, k0, f, queue, lane, count);
        const uint32_t kn = (k0 + 8 < n8) ? k0 + 8 : k0;  // last trip re-reads its own rows (stays in bounds)
        scan_rows_arrived(b0, b1, b2, b3);
        a0 = load_scan_row(rows, kn); a1 = load_scan_row(rows, kn + 1);
        a2 = load_scan_row(rows, kn + 2); a3 = load_scan_row(rows, kn + 3);
        filter_four(b0, b1, b2, b3, k0 + 4, f, queue, lane, count);
        if (__any(count > (uint32_t)(kQueueCap - 8))) drain_filtered<ROWS_IN_LDS>(spheres, planes_off, n_padded, queue, lane, count, r, a, tmin, closest, best_k);
    }
    for (uint32_t k = n8; k < n; k++) {
        const SphereScanRow g = load_scan_row(rows, k);
        const double s = filter_s(f, g.cx, g.cy, g.cz);
        const double q = filter_q(f, s, g.cx, g.cy, g.cz);
        if (q > g.k && !filter_behind(f, s, q, g.k)) {
            queue[count * 64u + lane] = (uint16_t)k;
            count++;
        }
        if (__any(count >= (uint32_t)kQueueCap)) drain_filtered<ROWS_IN_LDS>(spheres, planes_off, n_padded, queue, lane, count, r, a, tmin, closest, best_k);
    }
    drain_filtered<ROWS_IN_LDS>(spheres, planes_off, n_padded, queue, lane, count, r, a, tmin, closest, best_k);
    if (best_k == kNone) return false;
    best.t = closest;
    best.ref = make_ref(REF_SPHERE, best_k);
    best.obj = kNone;
    return true;
}

// Pixel-parallel scan: every live lane traces its own ray; sphere rows are wave-uniform (scalar path).
DEV bool scan_uniform(const DeviceScene &sc, uint16_t *queue, uint32_t lane, const Ray &r, double tmin, double tmax, HitInfo &best)
{
    const SphereGeom *__restrict__ spheres = sc.spheres;
    const uint32_t n = sc.n_spheres;
    const uint32_t n8 = n & ~7u;
    const double a = dot(r.d, r.d);
    double closest = tmax;
    uint32_t best_k = kNone, count = 0;
    // Eight rows per trip in two register sets (A, B): while one set is evaluated the other one's scalar loads are
    // in flight, and no set is ever copied into another.
    SphereGeom a0{}, a1{}, a2{}, a3{};
    if (n8) {
        a0 = load_sphere_row(spheres, 0); a1 = load_sphere_row(spheres, 1);
        a2 = load_sphere_row(spheres, 2); a3 = load_sphere_row(spheres, 3);
    }
    for (uint32_t k0 = 0; k0 < n8; k0 += 8) {
        rows_arrived(a0, a1, a2, a3);
        const SphereGeom b0 = load_sphere_row(spheres, k0 + 4), b1 = load_sphere_row(spheres, k0 + 5);
        const SphereGeom b2 = load_sphere_row(spheres, k0 + 6), b3 = load_sphere_row(spheres, k0 + 7);
        scan_four(a0, a1, a2, a3, k0, r, a, queue, lane, count);
        const uint32_t kn = (k0 + 8 < n8) ? k0 + 8 : k0;  // last trip re-reads its own rows (stays in bounds)
        rows_arrived(b0, b1, b2, b3);
        a0 = load_sphere_row(spheres, kn); a1 = load_sphere_row(spheres, kn + 1);
        a2 = load_sphere_row(spheres, kn + 2); a3 = load_sphere_row(spheres, kn + 3);
        scan_four(b0, b1, b2, b3, k0 + 4, r, a, queue, lane, count);
        if (__any(count > (uint32_t)(kQueueCap - 8))) drain_queue(spheres, queue, lane, count, r, a, tmin, closest, best_k);
    }
    for (uint32_t k = n8; k < n; k++) {
        scan_one(load_sphere_row(spheres, k), k, r, a, queue, lane, count);
        if (__any(count >= (uint32_t)kQueueCap)) drain_queue(spheres, queue, lane, count, r, a, tmin, closest, best_k);
    }
    drain_queue(spheres, queue, lane, count, r, a, tmin, closest, best_k);
    if (best_k == kNone) return false;
    best.t = closest;
    best.ref = make_ref(REF_SPHERE, best_k);
    best.obj = kNone;
    return true;
}

DEV double bcast(double x, int src_lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), src_lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), src_lane);
    return __hiloint2double(hi, lo);
}

// Sphere table as seen by the cooperative scan: four SoA planes in LDS (consecutive lanes read consecutive
// 8-byte words: conflict-free), or the global AoS table when it does not fit.
struct SphereView {
    // The LDS planes are addressed as byte offsets off the __shared__ symbol (plane(p, k)), never through pointers: a
    // pointer that may be LDS or global makes the compiler emit flat loads (19 of them in this kernel before), which are
    // slower than ds_read and tie up both wait counters.
    uint32_t planes_off;  // byte offset of plane 0 (cx); planes of n_padded doubles: cx, cy, cz, r2, K (filter_four's |c|^2 - r^2)
    double reach;         // DeviceScene::scan_reach
    bool exact;           // RT_FLAG_EXACT_SCAN: the reference's discriminant for every sphere, no filter
    const SphereGeom *global;
    uint32_t n, n_padded;
    bool in_lds;
    DEV double plane(uint32_t p, uint32_t k) const
    {
        return reinterpret_cast<const double *>(lds_raw + planes_off)[p * n_padded + k];
    }
};

// Wave-wide minimum of (t, k) pairs with DPP row operations (no LDS traffic); result valid in every lane.
// t > 0 always, so comparing (t, k) lexicographically picks the closest root, lowest sphere index on ties.
template <int CTRL, int ROW_MASK>
DEV void min_step(double &t, uint32_t &k)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(t), __double2loint(t), CTRL, ROW_MASK, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(t), __double2hiint(t), CTRL, ROW_MASK, 0xf, false);
    uint32_t ok = (uint32_t)__builtin_amdgcn_update_dpp((int)k, (int)k, CTRL, ROW_MASK, 0xf, false);
    double ot = __hiloint2double(hi, lo);
    bool take = (ot < t) || (ot == t && ok < k);
    t = take ? ot : t;
    k = take ? ok : k;
}
DEV void wave_min(double &t, uint32_t &k)
{
    min_step<0xB1, 0xf>(t, k);   // quad_perm [1,0,3,2]
    min_step<0x4E, 0xf>(t, k);   // quad_perm [2,3,0,1]
    min_step<0x114, 0xf>(t, k);  // row_shr:4
    min_step<0x118, 0xf>(t, k);  // row_shr:8   -> lane 15 of each row holds the row minimum
    min_step<0x142, 0xa>(t, k);  // row_bcast:15 into rows 1 and 3
    min_step<0x143, 0xc>(t, k);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave minimum
    t = bcast(t, 63);
    k = (uint32_t)__builtin_amdgcn_readlane((int)k, 63);
}

// Ray-cooperative scan: all 64 lanes work on ONE ray (lane `src`'s), lane l testing spheres l, l+64, ...;
// then a wave-wide min over (t, index).  Same winner as the sequential scan: that scan returns the smallest
// "first acceptable root" over all spheres with ties going to the lowest index, and a root that would have
// been rejected only because of an earlier closer hit loses the min here as well.
DEV void scan_cooperative(const SphereView &sv, uint32_t lane, unsigned long long todo, const Ray &ray, double tmin, double tmax,
                          HitInfo &best, bool &hit)
{
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        Ray r;
        r.o = mk(bcast(ray.o.x, src), bcast(ray.o.y, src), bcast(ray.o.z, src));
        r.d = mk(bcast(ray.d.x, src), bcast(ray.d.y, src), bcast(ray.d.z, src));
        const double a = dot(r.d, r.d);
        double bt = tmax;
        uint32_t bk = kNone;
        if (sv.in_lds && !sv.exact) {
            const ScanRay f = scan_ray(r, a, sv.reach);
            for (uint32_t base = 0; base < sv.n_padded; base += 256u) {
                double q[4];
                bool pass[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    uint32_t k = base + 64u * u + lane;
                    k = k < sv.n_padded ? k : sv.n_padded - 1u;
                    const double cx = sv.plane(0, k), cy = sv.plane(1, k), cz = sv.plane(2, k);
                    const double fs = filter_s(f, cx, cy, cz);
                    q[u] = filter_q(f, fs, cx, cy, cz);
                    pass[u] = q[u] > sv.plane(4, k);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t k = base + 64u * u + lane;
                    if (k < sv.n && pass[u]) {  // the reference's whole test for the survivors
                        double t;
                        if (sphere_test(r.o - mk(sv.plane(0, k), sv.plane(1, k), sv.plane(2, k)), r.d, a, sv.plane(3, k), tmin, bt, t)) {
                            bt = t;
                            bk = k;
                        }
                    }
                }
            }
        } else if (sv.in_lds) {
            for (uint32_t base = 0; base < sv.n_padded; base += 256u) {
                double b[4], c[4], disc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    // rows past n_padded are never read: clamp keeps the address inside the planes
                    uint32_t k = base + 64u * u + lane;
                    k = k < sv.n_padded ? k : sv.n_padded - 1u;
                    Vec oc = r.o - mk(sv.plane(0, k), sv.plane(1, k), sv.plane(2, k));
                    b[u] = dot(oc, r.d);
                    c[u] = dot(oc, oc) - sv.plane(3, k);
                    disc[u] = b[u] * b[u] - a * c[u];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t k = base + 64u * u + lane;
                    if (k < sv.n && disc[u] > 0.0 && !(b[u] > 0.0 && c[u] > 0.0)) {
                        double t;
                        if (sphere_roots(b[u], disc[u], a, tmin, bt, t)) {
                            bt = t;
                            bk = k;
                        }
                    }
                }
            }
        } else {
            for (uint32_t k = lane; k < sv.n; k += 64u) {
                SphereGeom g = sv.global[k];
                Vec oc = r.o - mk(g.cx, g.cy, g.cz);
                double b = dot(oc, r.d);
                double c = dot(oc, oc) - g.r2;
                double disc = b * b - a * c;
                if (disc > 0.0 && !(b > 0.0 && c > 0.0)) {
                    double t;
                    if (sphere_roots(b, disc, a, tmin, bt, t)) {
                        bt = t;
                        bk = k;
                    }
                }
            }
        }
        wave_min(bt, bk);
        if ((int)lane == src) {
            hit = bk != kNone;
            best.t = bt;
            best.ref = make_ref(REF_SPHERE, bk);
            best.obj = kNone;
        }
    }
}

// Lane exchange through the LDS crossbar (ds_bpermute: no memory traffic).
DEV int lane_read(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
DEV double lane_read(double x, int src_lane)
{
    int lo = lane_read(__double2loint(x), src_lane);
    int hi = lane_read(__double2hiint(x), src_lane);
    return __hiloint2double(hi, lo);
}
DEV void min_with_lane(double &t, uint32_t &k, int partner)
{
    double ot = lane_read(t, partner);
    uint32_t ok = (uint32_t)lane_read((int)k, partner);
    bool take = (ot < t) || (ot == t && ok < k);
    t = take ? ot : t;
    k = take ? ok : k;
}

// Ray-cooperative scan for a thin wave, several rays at a time: the L live rays are dealt to L groups of g = 64 / m
// lanes (m = L rounded up to a power of two), lane s of a group tests spheres s, s+g, s+2g, ...; one reduction over
// (t, index) per group serves all the rays at once.  Same winner as scan_cooperative (and as the sequential scan),
// at 1/L of its per-ray bookkeeping; against the pixel-parallel scan every lane does 1/g of the sphere tests.
DEV void scan_grouped(const SphereView &sv, uint32_t lane, unsigned long long todo, const Ray &ray, double tmin, double tmax,
                      HitInfo &best, bool &hit)
{
    const int L = __popcll(todo);
    int log2m = 0;
    while ((1 << log2m) < L) log2m++;
    const int log2g = 6 - log2m;
    const uint32_t g = 1u << log2g;
    const uint32_t q = lane >> log2g, s = lane & (g - 1u);  // my ray slot and my place in its group
    // owner of slot q = the q-th live lane
    int owner = (int)lane;
    {
        unsigned long long m = todo;
        for (int r = 0; r < L; r++) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            owner = q == (uint32_t)r ? src : owner;
        }
    }
    Ray r;
    r.o = mk(lane_read(ray.o.x, owner), lane_read(ray.o.y, owner), lane_read(ray.o.z, owner));
    r.d = mk(lane_read(ray.d.x, owner), lane_read(ray.d.y, owner), lane_read(ray.d.z, owner));
    const double a = dot(r.d, r.d);
    double bt = tmax;
    uint32_t bk = kNone;
    if (!sv.exact) {
        const ScanRay f = scan_ray(r, a, sv.reach);
        for (uint32_t base = 0; base < sv.n_padded; base += 4u * g) {
            bool pass[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                uint32_t k = base + g * u + s;
                k = k < sv.n_padded ? k : sv.n_padded - 1u;  // keeps the address inside the planes
                const double cx = sv.plane(0, k), cy = sv.plane(1, k), cz = sv.plane(2, k);
                const double fs = filter_s(f, cx, cy, cz);
                pass[u] = filter_q(f, fs, cx, cy, cz) > sv.plane(4, k);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t k = base + g * u + s;
                if (k < sv.n && pass[u]) {  // the reference's whole test for the survivors
                    double t;
                    if (sphere_test(r.o - mk(sv.plane(0, k), sv.plane(1, k), sv.plane(2, k)), r.d, a, sv.plane(3, k), tmin, bt, t)) {
                        bt = t;
                        bk = k;
                    }
                }
            }
        }
    } else
    for (uint32_t base = 0; base < sv.n_padded; base += 4u * g) {
        double b[4], c[4], disc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint32_t k = base + g * u + s;
            k = k < sv.n_padded ? k : sv.n_padded - 1u;  // keeps the address inside the planes
            Vec oc = r.o - mk(sv.plane(0, k), sv.plane(1, k), sv.plane(2, k));
            b[u] = dot(oc, r.d);
            c[u] = dot(oc, oc) - sv.plane(3, k);
            disc[u] = b[u] * b[u] - a * c[u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t k = base + g * u + s;
            if (k < sv.n && disc[u] > 0.0 && !(b[u] > 0.0 && c[u] > 0.0)) {
                double t;
                if (sphere_roots(b[u], disc[u], a, tmin, bt, t)) {
                    bt = t;
                    bk = k;
                }
            }
        }
    }
    // minimum over the group: partners at distance 1, 2 (quad permutes), 4 (half-row mirror), 8 (row mirror), then 16, 32
    if (log2g >= 1) min_step<0xB1, 0xf>(bt, bk);   // quad_perm [1,0,3,2]
    if (log2g >= 2) min_step<0x4E, 0xf>(bt, bk);   // quad_perm [2,3,0,1]
    if (log2g >= 3) min_step<0x141, 0xf>(bt, bk);  // row_half_mirror
    if (log2g >= 4) min_step<0x140, 0xf>(bt, bk);  // row_mirror
    if (log2g >= 5) min_with_lane(bt, bk, (int)lane ^ 16);
    if (log2g >= 6) min_with_lane(bt, bk, (int)lane ^ 32);
    // back to the owners: live lane number r reads the first lane of group r
    const int rank = __popcll(todo & ((1ull << lane) - 1ull));
    const double rt = lane_read(bt, rank << log2g);
    const uint32_t rk = (uint32_t)lane_read((int)bk, rank << log2g);
    if ((todo >> lane) & 1ull) {
        hit = rk != kNone;
        best.t = rt;
        best.ref = make_ref(REF_SPHERE, rk);
        best.obj = kNone;
    }
}

// ---- lanes-per-ray scan of a list world (R/HittableList.h:39-57) ---------------------------------------------------------
// A pixel's samples are one sequential chain (one RNG stream), so when a launch has fewer pixels than the device has lanes
// -- a small frame, one rank's stripes of a frame -- nothing shortens the frame but a shorter chain per ray.  With
// pixels_per_wave = 64 / g the pixels sit in the wave's first 64 / g lanes and the g lanes of group q share the ray of lane
// q: lane s of the group tests leaves s, s + g, ... with its own running closest hit, then one reduction per group picks
// the hit HittableList::Hit returns.  Which one that is when two leaves answer the SAME t depends on the kinds: a sphere
// is accepted for t < closest only (R/Sphere.h:38,50), a quad -- hence a box face, an instanced box -- for t <= closest
// (R/Quad.h:59-64 rejects t > tMax only).  Walking the list in order, the first leaf that reaches the final t sets it, a
// later quad at that t replaces it, a later sphere does not: the winner is the LAST quad among the tied leaves if there is
// one, else the FIRST sphere.  `key` orders exactly that (quads above spheres, later quads and earlier spheres higher),
// and every lane's own sub-sequence obeys the same rule by running the reference's tests in order.  A leaf's answer under
// a looser bound than the list would have given it is the same answer or one that loses the reduction (no leaf draws
// random numbers in these kernels: T::MEDIA is false), so the frame is the sequential scan's bit for bit.
template <int CTRL, int ROW_MASK>
DEV void tie_step(double &t, uint32_t &key)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(t), __double2loint(t), CTRL, ROW_MASK, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(t), __double2hiint(t), CTRL, ROW_MASK, 0xf, false);
    uint32_t ok = (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, CTRL, ROW_MASK, 0xf, false);
    double ot = __hiloint2double(hi, lo);
    bool take = (ot < t) || (ot == t && ok > key);
    t = take ? ot : t;
    key = take ? ok : key;
}
DEV void tie_with_lane(double &t, uint32_t &key, int partner)
{
    double ot = lane_read(t, partner);
    uint32_t ok = (uint32_t)lane_read((int)key, partner);
    bool take = (ot < t) || (ot == t && ok > key);
    t = take ? ot : t;
    key = take ? ok : key;
}

template <class T>
DEV bool leaf_test(const DeviceScene &sc, uint32_t ref, const Ray &r, double a, double tmin, double tmax, HitInfo &best, Xorwow &rng PH_ARG);

template <class T>
DEV void scan_leaves_grouped(const DeviceScene &sc, uint32_t lane, int log2g, unsigned long long todo, const Ray &ray, double tmin, double tmax,
                             HitInfo &best, bool &hit, Xorwow &rng PH_ARG)
{
    const uint32_t g = 1u << log2g;
    const uint32_t q = lane >> log2g, s = lane & (g - 1u);  // group q serves the pixel of lane q; my place in the group
    Ray r;
    r.o = mk(lane_read(ray.o.x, (int)q), lane_read(ray.o.y, (int)q), lane_read(ray.o.z, (int)q));
    r.d = mk(lane_read(ray.d.x, (int)q), lane_read(ray.d.y, (int)q), lane_read(ray.d.z, (int)q));
    r.tm = lane_read(ray.tm, (int)q);
    const bool wanted = ((todo >> q) & 1ull) != 0;
    const double a = dot(r.d, r.d);
    const uint32_t n = sc.n_world_items;
    HitInfo mine;
    mine.t = 0.0;
    mine.ref = kNone;
    mine.obj = kNone;
    double closest = tmax;
    uint32_t k_best = 0;
    bool any = false;
    if (wanted) {
        for (uint32_t k = s; k < n; k += g) {
            const uint32_t ref = ((const RT_CONST uint32_t *)(uintptr_t)sc.world_items)[k];
            if (leaf_test<T>(sc, ref, r, a, tmin, closest, mine, rng PH_PASS)) {
                any = true;
                closest = mine.t;
                k_best = k;
            }
        }
    }
    double wt = any ? closest : __builtin_inf();
    const uint32_t my_key = !any ? 0u : ((mine.ref >> kRefShift) == REF_QUAD ? (0x80000000u | k_best) : (0x7FFFFFFFu - k_best));
    uint32_t wkey = my_key;
    if (log2g >= 1) tie_step<0xB1, 0xf>(wt, wkey);   // quad_perm [1,0,3,2]
    if (log2g >= 2) tie_step<0x4E, 0xf>(wt, wkey);   // quad_perm [2,3,0,1]
    if (log2g >= 3) tie_step<0x141, 0xf>(wt, wkey);  // row_half_mirror
    if (log2g >= 4) tie_step<0x140, 0xf>(wt, wkey);  // row_mirror
    if (log2g >= 5) tie_with_lane(wt, wkey, (int)lane ^ 16);
    if (log2g >= 6) tie_with_lane(wt, wkey, (int)lane ^ 32);
    // every lane of a group now holds the group's (t, key); the lane whose own answer that is hands its record to the pixel's lane
    const unsigned long long holders = __ballot(any && my_key == wkey);
    const unsigned long long group_bits = log2g >= 6 ? holders : ((holders >> (lane << log2g)) & ((1ull << g) - 1ull));  // as owner: group `lane`
    const int src = (int)(lane << log2g) + (group_bits ? __ffsll((long long)group_bits) - 1 : 0);
    const double rt = lane_read(wt, src & 63);
    const uint32_t rref = (uint32_t)lane_read((int)mine.ref, src & 63);
    const uint32_t robj = (uint32_t)lane_read((int)mine.obj, src & 63);
    if (lane < (64u >> log2g) && ((todo >> lane) & 1ull)) {
        hit = group_bits != 0;
        best.t = rt;
        best.ref = rref;
        best.obj = robj;
    }
}

// The same grouped scan for a BVH world of spheres / unit-time moving spheres (config C3), for thin waves once the pixel
// queue has run dry.  A pixel's samples are one sequential chain, so a frame ends with a few long pixels (glass: up to
// max_depth rays per sample); walked, each of their rays costs ~36 dependent node visits with most of the wave idle.
// Scanned, the L live rays are dealt to groups of 64 / m lanes, every lane tests its share of ALL leaves (seven coalesced
// plane reads per row) and one reduction per group picks the hit -- the same hit as the walk's: no leaf draws random
// numbers, so the walk returns the closest acceptable root over all leaves (the reference's BVH = list invariant, Docs
// 2-3 BVH :733,:772), ties going to the lower leaf index here.  Centre and roots are computed exactly as prim_test does.
DEV void scan_grouped_ms(const DeviceScene &sc, uint32_t lane, unsigned long long todo, const Ray &ray, double tmin, double tmax,
                         HitInfo &best, bool &hit)
{
    const uint32_t n = sc.n_world_items, np = sc.ms_padded;
    const double *__restrict__ pl = sc.ms_planes;
    const int L = __popcll(todo);
    int log2m = 0;
    while ((1 << log2m) < L) log2m++;
    const int log2g = 6 - log2m;
    const uint32_t g = 1u << log2g;
    const uint32_t q = lane >> log2g, s = lane & (g - 1u);  // my ray slot and my place in its group
    int owner = (int)lane;
    {
        unsigned long long m = todo;
        for (int r = 0; r < L; r++) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            owner = q == (uint32_t)r ? src : owner;
        }
    }
    Ray r;
    r.o = mk(lane_read(ray.o.x, owner), lane_read(ray.o.y, owner), lane_read(ray.o.z, owner));
    r.d = mk(lane_read(ray.d.x, owner), lane_read(ray.d.y, owner), lane_read(ray.d.z, owner));
    r.tm = lane_read(ray.tm, owner);
    const double a = dot(r.d, r.d);
    double bt = tmax;
    uint32_t bk = kNone;
    for (uint32_t base = 0; base < np; base += 4u * g) {
        double b[4], c[4], disc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint32_t k = base + g * u + s;
            k = k < np ? k : np - 1u;  // keeps the address inside the planes
            const Vec c0 = mk(pl[k], pl[np + k], pl[2u * np + k]);
            const Vec dc = mk(pl[3u * np + k], pl[4u * np + k], pl[5u * np + k]);
            const Vec centre = c0 + r.tm * dc;  // msphere_center with unit time (a static sphere: dc = 0)
            Vec oc = r.o - centre;
            b[u] = dot(oc, r.d);
            c[u] = dot(oc, oc) - pl[6u * np + k];
            disc[u] = b[u] * b[u] - a * c[u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t k = base + g * u + s;
            if (k < n && disc[u] > 0.0 && !(tmin >= 0.0 && b[u] > 0.0 && c[u] > 0.0)) {  // see sphere_test
                double t;
                if (sphere_roots(b[u], disc[u], a, tmin, bt, t)) {
                    bt = t;
                    bk = k;
                }
            }
        }
    }
    if (log2g >= 1) min_step<0xB1, 0xf>(bt, bk);   // quad_perm [1,0,3,2]
    if (log2g >= 2) min_step<0x4E, 0xf>(bt, bk);   // quad_perm [2,3,0,1]
    if (log2g >= 3) min_step<0x141, 0xf>(bt, bk);  // row_half_mirror
    if (log2g >= 4) min_step<0x140, 0xf>(bt, bk);  // row_mirror
    if (log2g >= 5) min_with_lane(bt, bk, (int)lane ^ 16);
    if (log2g >= 6) min_with_lane(bt, bk, (int)lane ^ 32);
    const int rank = __popcll(todo & ((1ull << lane) - 1ull));
    const double rt = lane_read(bt, rank << log2g);
    const uint32_t rk = (uint32_t)lane_read((int)bk, rank << log2g);
    if ((todo >> lane) & 1ull) {
        hit = rk != kNone;
        best.t = rt;
        best.ref = hit ? sc.world_items[rk] : kNone;
        best.obj = kNone;
    }
}

// ------------------------------------------------------------------------------------------------
// hit record, built once per bounce from (t, primitive)
// ------------------------------------------------------------------------------------------------
DEV void sphere_uv(Vec on, double &u, double &v)  // R/Sphere.h:74-81
{
    const double pi = 3.1415926535897932385;
    double theta = acos(-on.y);
    double phi = atan2(-on.z, on.x) + pi;
    u = phi / (2.0 * pi);
    v = theta / pi;
}

DEV void face(Surface &s, const Ray &r, Vec outward)  // R/Hittable.h:26-30
{
    s.front = dot(r.d, outward) < 0.0;
    s.n = s.front ? outward : -outward;
}

template <class T>
DEV Surface make_surface(const DeviceScene &sc, const Ray &r, const HitInfo &h)
{
    Surface s;
    s.u = 0.0;
    s.v = 0.0;
    uint32_t tag = h.ref >> kRefShift, idx = h.ref & kRefIndexMask;
    // the transforms between the world and the space the hit was found in (outermost first), if any
    uint32_t xf_first = 0, xf_count = 0;
    Ray lr = r;
    if constexpr (T::COMPOSITE) {
        if (h.obj != kNone) {
            if (T::NESTED && (h.obj & kTreeObjBit)) {  // inside a tree: the winning node's chain
                const TreeNodeRec n = sc.tree_nodes[h.obj & ~kTreeObjBit];
                xf_first = n.chain_first;
                xf_count = n.chain_count;
            } else if (tag != REF_MEDIUM) {
                const ObjectRec o = get_object(sc, h.obj);
                xf_first = o.xf_first;
                xf_count = o.xf_count;
            }
            lr = chain_ray(sc, xf_first, xf_count, r);
        }
    }
    bool is_medium = false;
    if constexpr (T::MEDIA) is_medium = tag == REF_MEDIUM;
    if (is_medium) {  // R/ConstantMedium.h:86-91
        s.p = at(lr, h.t);
        s.n = mk(1, 0, 0);
        s.front = true;
        s.mat = get_medium(sc, idx).phase_mat;
    } else
    if (T::WORLD != 2 && tag == REF_QUAD) {  // R/Quad.h:86-96
        s.p = at(lr, h.t);
        QuadGeom q = sc.quads[idx];
        s.mat = sc.quad_mat[idx];
        face(s, lr, mk(q.nx, q.ny, q.nz));
        if constexpr (T::RICH) {
            if (material_needs_uv(sc, s.mat)) {
                Vec ph = s.p - mk(q.qx, q.qy, q.qz);
                Vec w = mk(q.wx, q.wy, q.wz);
                s.u = dot(w, cross(ph, mk(q.vx, q.vy, q.vz)));
                s.v = dot(w, cross(mk(q.ux, q.uy, q.uz), ph));
            }
        }
    } else {  // R/Sphere.h:40-46
        s.p = at(lr, h.t);
        Vec c;
        SphereAux aux;
        if (T::WORLD == 2 || tag == REF_SPHERE) {
            SphereGeom g = T::COMPOSITE ? get_sphere(sc, idx) : (T::FAST ? get_sphere_typed(sc, idx) : sc.spheres[idx]);
            c = mk(g.cx, g.cy, g.cz);
            aux = T::FAST ? get_sphere_aux(sc, idx) : sc.sphere_aux[idx];
        } else {
            c = msphere_center(T::FAST ? get_msphere(sc, idx) : sc.mspheres[idx], lr.tm, (sc.flags & SCENE_MS_UNIT_TIME) != 0);
            aux = T::FAST ? get_msphere_aux(sc, idx) : sc.msphere_aux[idx];
        }
        Vec on = aux.inv_r * (s.p - c);
        face(s, lr, on);
        s.mat = aux.mat;
        if constexpr (T::RICH) {
            if (material_needs_uv(sc, s.mat)) sphere_uv(on, s.u, s.v);
        }
    }
    if constexpr (T::COMPOSITE) {
        if (h.obj != kNone) {  // back to world space, innermost transform first (R/Instance.h:53,137-147)
            for (uint32_t k = xf_count; k-- > 0;) {
                Xform x = get_xform(sc, xf_first + k);
                if (x.kind == XF_TRANSLATE) {
                    s.p = s.p + mk(x.a, x.b, x.c);
                } else {
                    double st = x.a, ct = x.b;
                    s.p = mk((ct * s.p.x) + (st * s.p.z), s.p.y, (-st * s.p.x) + (ct * s.p.z));
                    s.n = mk((ct * s.n.x) + (st * s.n.z), s.n.y, (-st * s.n.x) + (ct * s.n.z));
                }
            }
        }
    }
    return s;
}

// ------------------------------------------------------------------------------------------------
// textures and materials
// ------------------------------------------------------------------------------------------------
DEV double perlin_noise(const PerlinRec *pn, Vec p)  // R/Perlin.h:38-60,120-139
{
    double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    double uu = u * u * (3.0 - 2.0 * u), vv = v * v * (3.0 - 2.0 * v), ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
    // corners one after another (same summation order as the reference; keeps the register footprint small)
#pragma unroll 1
    for (int a = 0; a < 2; a++)
#pragma unroll 1
        for (int b = 0; b < 2; b++)
#pragma unroll 1
            for (int c = 0; c < 2; c++) {
                int idx = pn->perm_x[(i + a) & 255] ^ pn->perm_y[(j + b) & 255] ^ pn->perm_z[(k + c) & 255];
                Vec g = mk(pn->vec[idx][0], pn->vec[idx][1], pn->vec[idx][2]);
                Vec wv = mk(u - a, v - b, w - c);
                accum += (a * uu + (1 - a) * (1 - uu)) * (b * vv + (1 - b) * (1 - vv)) * (c * ww + (1 - c) * (1 - ww)) * dot(g, wv);
            }
    return accum;
}

// The same from a table staged in LDS (layout of PerlinRec at byte offset `off`).  The global version above walks the
// eight corners one dependent load chain after another (56 chains of two L2 round trips per marble lookup: one marble
// lane held its whole wave for ~50k cycles); here the six permutation entries and the eight gradient rows are
// independent LDS reads, summed in the reference's corner order.
DEV double perlin_noise_lds(uint32_t off, Vec p)  // R/Perlin.h:38-60,120-139
{
    const double *vec = reinterpret_cast<const double *>(lds_raw + off);
    const int32_t *perm = reinterpret_cast<const int32_t *>(lds_raw + off + 256u * 3u * (uint32_t)sizeof(double));
    double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    double uu = u * u * (3.0 - 2.0 * u), vv = v * v * (3.0 - 2.0 * v), ww = w * w * (3.0 - 2.0 * w);
    const int px[2] = {perm[i & 255], perm[(i + 1) & 255]};
    const int py[2] = {perm[256 + (j & 255)], perm[256 + ((j + 1) & 255)]};
    const int pz[2] = {perm[512 + (k & 255)], perm[512 + ((k + 1) & 255)]};
    double accum = 0.0;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int idx = px[a] ^ py[b] ^ pz[c];
                Vec g = mk(vec[idx * 3 + 0], vec[idx * 3 + 1], vec[idx * 3 + 2]);
                Vec wv = mk(u - a, v - b, w - c);
                accum += (a * uu + (1 - a) * (1 - uu)) * (b * vv + (1 - b) * (1 - vv)) * (c * ww + (1 - c) * (1 - ww)) * dot(g, wv);
            }
    return accum;
}

DEV double perlin_turb(const DeviceScene &sc, uint32_t table, Vec p, int depth)  // R/Perlin.h:63-78
{
    double accum = 0.0, weight = 1.0;
    const bool in_lds = sc.lds_perlin != kNone;
    for (int i = 0; i < depth; i++) {
        accum += weight * (in_lds ? perlin_noise_lds(sc.lds_perlin + table * (uint32_t)sizeof(PerlinRec), p) : perlin_noise(sc.perlin + table, p));
        weight *= 0.5;
        p = 2.0 * p;
    }
    return fabs(accum);
}

template <class T>
DEV Vec texture_value(const DeviceScene &sc, uint32_t ti, double u, double v, Vec p)
{
    TextureRec t = sc.textures[ti];
    while (t.kind == TEX_CHECKER) {  // R/Texture.h:70-81
        int xi = (int)floor(t.s * p.x), yi = (int)floor(t.s * p.y), zi = (int)floor(t.s * p.z);
        bool even = ((xi + yi + zi) % 2) == 0;
        t = sc.textures[even ? t.a : t.b_];
    }
    if constexpr (T::RICH) {
        if (t.kind == TEX_IMAGE) {  // R/Texture.h:110-133
            ImageRec im = sc.images[t.a];
            if (im.height <= 0) return mk(0.0, 1.0, 1.0);
            u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            v = 1.0 - v;
            int i = (int)(u * im.width), j = (int)(v * im.height);
            if (i >= im.width) i = im.width - 1;
            if (j >= im.height) j = im.height - 1;
            const unsigned char *px = sc.image_bytes + im.offset + ((size_t)j * im.width + i) * 3;
            double cs = 1.0 / 255.0;
            return mk(cs * px[0], cs * px[1], cs * px[2]);
        }
        if (t.kind == TEX_NOISE) {  // R/Texture.h:159-165: marble
            double sv = 1.0 + sin(t.s * p.z + 10.0 * perlin_turb(sc, t.a, p, 7));
            return sv * mk(0.5, 0.5, 0.5);
        }
    }
    return mk(t.r, t.g, t.b);  // TEX_SOLID, R/Texture.h:48-51
}

// One material row, read field by field from wherever the table lives: the LDS copy (small tables, launcher's choice)
// or the global table.  Which one is wave-uniform, so each read is one scalar branch and one load.
struct MatView {
    const MaterialRec *global;
    uint32_t lds_off;  // byte offset of the row in LDS, or kNone
    // The LDS side is read through a pointer typed for the LDS address space: two loads through generic pointers get merged
    // into one flat load through a selected pointer (slower than ds_read, and it ties up both wait counters).
    DEV double f64(uint32_t off) const
    {
        if (lds_off != kNone) return *(const RT_LDS double *)(lds_raw + lds_off + off);
        return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(global) + off);
    }
    DEV uint32_t u32(uint32_t off) const
    {
        if (lds_off != kNone) return *(const RT_LDS uint32_t *)(lds_raw + lds_off + off);
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(global) + off);
    }
    DEV Vec vec(uint32_t off) const { return mk(f64(off), f64(off + 8u), f64(off + 16u)); }
};
#define MAT_OFF(field) ((uint32_t)offsetof(MaterialRec, field))
// MAY_BE_STAGED = false (kernels that never stage materials): the LDS side folds away at compile time -- left in, the
// compiler merges the two loads of every field into one flat load through a selected pointer
template <bool MAY_BE_STAGED>
DEV MatView material_view(const DeviceScene &sc, uint32_t i)
{
    if constexpr (!MAY_BE_STAGED) return MatView{sc.materials + i, kNone};
    return MatView{sc.materials + i, sc.lds_materials != kNone ? sc.lds_materials + i * (uint32_t)sizeof(MaterialRec) : kNone};
}

// Texture of a material: host-resolved solid / checker-of-solids without touching the texture table.
template <class T>
DEV Vec material_texture(const DeviceScene &sc, const MatView &m, uint32_t tex_inline, double u, double v, Vec p)
{
    if (tex_inline == 1) return m.vec(MAT_OFF(even));
    if (tex_inline == 2) {  // R/Texture.h:70-81
        const double inv_scale = m.f64(MAT_OFF(inv_scale));
        int xi = (int)floor(inv_scale * p.x), yi = (int)floor(inv_scale * p.y), zi = (int)floor(inv_scale * p.z);
        bool even = ((xi + yi + zi) % 2) == 0;
        return m.vec(even ? MAT_OFF(even) : MAT_OFF(odd));
    }
    if constexpr (T::RICH) return texture_value<T>(sc, m.u32(MAT_OFF(tex)), u, v, p);
    return mk(0.0, 0.0, 0.0);  // unreachable: scenes with table-walking textures run the RICH instantiation
}

// pow(x, 5.0) of Schlick's approximation (R/Dielectric.h:61-67) by repeated multiplication: within 1.5 ulp of the
// correctly rounded value, i.e. the same accuracy class as any libm's pow (the reference's is CUDA's, the oracle's
// is glibc's, neither is bit-reproducible here).  The value only feeds the comparison `reflectance > uniform`.
// The library pow() costs ~45 VGPRs of kernel-wide register budget -- a whole wave per SIMD of occupancy.
DEV double pow5(double x)
{
    double x2 = x * x;
    double x4 = x2 * x2;
    return x4 * x;
}

DEV Vec random_in_unit_sphere(Xorwow &rng)  // R/Material.h:14-24
{
    Vec p;
    do {
        double a = (double)xorwow_uniform(rng);
        double b = (double)xorwow_uniform(rng);
        double c = (double)xorwow_uniform(rng);
        p = 2.0 * mk(a, b, c) - mk(1.0, 1.0, 1.0);
    } while (length_sq(p) >= 1.0);
    return p;
}

// Emitted + Scatter (R/kernel.cu:82-94).  Returns false when the path ends here.
// The in-sphere sample (Lambertian, Metal, Isotropic) and the unit direction (Metal, Dielectric) are evaluated
// once, ahead of the material switch, for the lanes whose material uses them: in each of those Scatter
// functions the sample is the first thing drawn from the RNG, so the per-pixel draw order is unchanged.
template <class T>
DEV bool shade(const DeviceScene &sc, const Surface &s, Ray &ray, Vec &throughput, Vec &accumulated, Xorwow &rng)
{
    // only the fields a material kind needs are loaded (the row is 112 bytes)
    const MatView mp = material_view<(T::COMPOSITE && T::WORLD == 0) || T::FAST>(sc, s.mat);
    struct { uint32_t kind, tex_inline; } m = {mp.u32(MAT_OFF(kind)), mp.u32(MAT_OFF(tex_inline))};
    Vec atten;
    Ray out;
    out.o = s.p;
    out.tm = ray.tm;
    Vec rs = mk(0.0, 0.0, 0.0), ud = mk(0.0, 0.0, 0.0);
    if (m.kind == MAT_LAMBERTIAN || m.kind == MAT_METAL || m.kind == MAT_ISOTROPIC) rs = random_in_unit_sphere(rng);
    if (m.kind == MAT_METAL || m.kind == MAT_DIELECTRIC) ud = unit(ray.d);
    switch (m.kind) {
    case MAT_DIFFUSE_LIGHT:  // R/Material.h:114-127: emits on both sides, never scatters
        accumulated = accumulated + throughput * material_texture<T>(sc, mp, m.tex_inline, s.u, s.v, s.p);
        return false;
    case MAT_LAMBERTIAN: {  // R/Material.h:67-82
        Vec dir = s.n + rs;
        if (fabs(dir.x) < 1e-8 && fabs(dir.y) < 1e-8 && fabs(dir.z) < 1e-8) dir = s.n;
        out.d = dir;
        atten = material_texture<T>(sc, mp, m.tex_inline, s.u, s.v, s.p);
        break;
    }
    case MAT_METAL: {  // R/Metal.h:18-30
        Vec refl = reflect(ud, s.n);
        out.d = refl + mp.f64(MAT_OFF(p)) * rs;
        atten = mp.vec(MAT_OFF(r));
        if (!(dot(out.d, s.n) > 0.0)) return false;
        break;
    }
    case MAT_DIELECTRIC: {  // R/Dielectric.h:18-68
        atten = mk(1.0, 1.0, 1.0);
        const double ior = mp.f64(MAT_OFF(p));
        double ratio = s.front ? (1.0 / ior) : ior;
        double ct = fmin(dot(-ud, s.n), 1.0);
        double st = sqrt(1.0 - ct * ct);
        bool reflect_it = ratio * st > 1.0;
        if (!reflect_it) {
            double r0 = (1.0 - ratio) / (1.0 + ratio);
            r0 = r0 * r0;
            double refl = r0 + (1.0 - r0) * pow5(1.0 - ct);
            reflect_it = refl > (double)xorwow_uniform(rng);
        }
        out.d = reflect_it ? reflect(ud, s.n) : refract(ud, s.n, ratio);
        break;
    }
    default: {  // MAT_ISOTROPIC, R/Material.h:152-163
        out.d = unit(rs);
        atten = material_texture<T>(sc, mp, m.tex_inline, s.u, s.v, s.p);
        break;
    }
    }
    throughput = throughput * atten;
    ray = out;
    return true;
}

// Camera::GetRay (R/Camera.h:76-85) behind the pixel jitter of Render (R/kernel.cu:140-142).
DEV Vec load3c(const RT_CONST double *p) { return mk(p[0], p[1], p[2]); }

DEV Ray camera_ray(const CameraRec *cam_generic, int i, int j, int width, int height, Xorwow &rng)
{
    const RT_CONST CameraRec *cam = (const RT_CONST CameraRec *)(uintptr_t)cam_generic;
    double u = (double)((float)i + xorwow_uniform(rng)) / (double)width;   // int + float adds in fp32
    double v = (double)((float)j + xorwow_uniform(rng)) / (double)height;
    Vec p;
    do {
        double a = (double)xorwow_uniform(rng);
        double b = (double)xorwow_uniform(rng);
        p = 2.0 * mk(a, b, 0.0) - mk(1.0, 1.0, 0.0);
    } while (dot(p, p) >= 1.0);
    Vec rd = cam->lens_radius * p;
    Vec cu = load3c(cam->u), cv = load3c(cam->v);
    Vec offset = rd.x * cu + rd.y * cv;
    double tm = cam->time0 + (double)xorwow_uniform(rng) * (cam->time1 - cam->time0);
    Vec origin = load3c(cam->origin);
    Ray r;
    r.o = origin + offset;
    r.d = load3c(cam->llc) + u * load3c(cam->horizontal) + v * load3c(cam->vertical) - origin - offset;
    r.tm = tm;
    return r;
}

// Row of the full frame that local row `lr` of this rank maps to (stripes dealt round-robin).
DEV int owned_row(int lr, int stripe, int rank, int world)
{
    return ((lr / stripe) * world + rank) * stripe + (lr % stripe);
}

} // namespace

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
template <int STRICT>
__global__ __launch_bounds__(256) void seed_kernel(SeedArgs a)
{
    uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= a.n_pixels) return;
    int lr = (int)(local / (uint32_t)a.width), i = (int)(local % (uint32_t)a.width);
    int j = owned_row(lr, a.stripe_rows, a.rank, a.world_size);
    uint64_t sequence = (uint64_t)((int64_t)j * a.width + i);  // pixelIndex, R/kernel.cu:117-118
    Xorwow s = a.base;
    xorwow_skip_sequences(a.jump_table, sequence, s);
    a.state[0 * (size_t)a.n_pixels + local] = s.d;
    a.state[1 * (size_t)a.n_pixels + local] = s.v0;
    a.state[2 * (size_t)a.n_pixels + local] = s.v1;
    a.state[3 * (size_t)a.n_pixels + local] = s.v2;
    a.state[4 * (size_t)a.n_pixels + local] = s.v3;
    a.state[5 * (size_t)a.n_pixels + local] = s.v4;
}


#if RT_PHASES
DEV void ph_flush(const RenderArgs &a, PhaseSums &ph, unsigned long long ph_start, uint32_t lane)
{
    for (int k = 4; k < 15; k++) {
        if (k == 12 || k == 13) continue;  // wave-level slots
        for (int off = 32; off > 0; off >>= 1) {
            ph.t[k] += __shfl_down(ph.t[k], off, 64);
            ph.l[k] += __shfl_down(ph.l[k], off, 64);
            ph.n[k] += __shfl_down(ph.n[k], off, 64);
        }
        ph.t[k] >>= 10;
        ph.l[k] >>= 10;
        ph.n[k] >>= 10;
    }
    if (lane == 0) {
        for (int k = 0; k < 24; k++) {
            atomicAdd(a.ray_counter + 32 + k, ph.t[k]);
            atomicAdd(a.ray_counter + 64 + k, ph.l[k]);
            atomicAdd(a.ray_counter + 96 + k, ph.n[k]);
        }
        atomicAdd(a.ray_counter + 7, __builtin_readcyclecounter() - ph_start);
    }
}
#endif
// Parked path state (Traits::PARK): twelve 8-byte planes of BLOCK entries behind the staged tables, one entry per thread.
template <int BLOCK>
DEV void park_put(uint32_t off, int k, double v) { *(RT_LDS double *)(lds_raw + off + ((uint32_t)k * (uint32_t)BLOCK + threadIdx.x) * 8u) = v; }
template <int BLOCK>
DEV double park_get(uint32_t off, int k) { return *(const RT_LDS double *)(lds_raw + off + ((uint32_t)k * (uint32_t)BLOCK + threadIdx.x) * 8u); }
template <int BLOCK>
DEV void park_put_vec(uint32_t off, int k, Vec v) { park_put<BLOCK>(off, k, v.x); park_put<BLOCK>(off, k + 1, v.y); park_put<BLOCK>(off, k + 2, v.z); }
template <int BLOCK>
DEV Vec park_get_vec(uint32_t off, int k) { return mk(park_get<BLOCK>(off, k), park_get<BLOCK>(off, k + 1), park_get<BLOCK>(off, k + 2)); }
template <int BLOCK>
DEV void park_put_rng(uint32_t off, const Xorwow &g)
{
    RT_LDS uint32_t *p = (RT_LDS uint32_t *)(lds_raw + off + (9u * (uint32_t)BLOCK) * 8u);
    p[0 * BLOCK + threadIdx.x] = g.d;  p[1 * BLOCK + threadIdx.x] = g.v0; p[2 * BLOCK + threadIdx.x] = g.v1;
    p[3 * BLOCK + threadIdx.x] = g.v2; p[4 * BLOCK + threadIdx.x] = g.v3; p[5 * BLOCK + threadIdx.x] = g.v4;
}
template <int BLOCK>
DEV void park_get_rng(uint32_t off, Xorwow &g)
{
    const RT_LDS uint32_t *p = (const RT_LDS uint32_t *)(lds_raw + off + (9u * (uint32_t)BLOCK) * 8u);
    g.d = p[0 * BLOCK + threadIdx.x];  g.v0 = p[1 * BLOCK + threadIdx.x]; g.v1 = p[2 * BLOCK + threadIdx.x];
    g.v2 = p[3 * BLOCK + threadIdx.x]; g.v3 = p[4 * BLOCK + threadIdx.x]; g.v4 = p[5 * BLOCK + threadIdx.x];
}
// ... and five 4-byte planes behind them: pixel column, row, index in the rank's buffer, sample number, rays of the pixel so far
template <int BLOCK>
DEV void park_put_int(uint32_t off, int k, uint32_t v) { *(RT_LDS uint32_t *)(lds_raw + off + (24u + (uint32_t)k) * (uint32_t)BLOCK * 4u + threadIdx.x * 4u) = v; }
template <int BLOCK>
DEV uint32_t park_get_int(uint32_t off, int k) { return *(const RT_LDS uint32_t *)(lds_raw + off + (24u + (uint32_t)k) * (uint32_t)BLOCK * 4u + threadIdx.x * 4u); }
constexpr size_t kParkBytesPerThread = 12 * 8 + 5 * 4;  // col, throughput, accumulated: 9 doubles; RNG: 6 words; 5 counters
template <int STRICT, class T>
__global__ __launch_bounds__(T::BLOCK, T::MIN_WAVES) void render_kernel(DeviceScene sc, RenderArgs a)
{
    // What the launcher never stages for this instantiation is said here at compile time (launch_one places tables for
    // BVH worlds only, the sphere / material rows of primitive worlds only for the library-tree kernel, the big tables of
    // composite worlds only for the 768-thread workgroup): the LDS side of those row accessors folds away, and a
    // wave-uniform row of a list kernel feeds the vector instructions straight from the scalar registers it was loaded
    // into instead of being copied into vector registers to meet the other path (C4 2268 -> 2472 Msamples/s).
    if constexpr (T::WORLD != 0 || (!T::COMPOSITE && !T::FAST)) {
        sc.lds_quad_aa = sc.lds_boxes = sc.lds_objects = sc.lds_xforms = sc.lds_media = sc.lds_materials = sc.lds_perlin = kNone;
        sc.lds_spheres_tab = sc.lds_group_boxes = sc.lds_mspheres = sc.lds_msphere_aux = sc.lds_sphere_aux = kNone;
    } else if constexpr (T::FAST) {
        sc.lds_quad_aa = sc.lds_boxes = sc.lds_objects = sc.lds_xforms = sc.lds_media = sc.lds_perlin = sc.lds_group_boxes = kNone;
#ifndef RT_NO_ASSUME
        if constexpr (T::BLOCK >= RT_BIG_BLOCK) {  // launched only with all of its rows staged (launch_one)
            __builtin_assume(sc.lds_mspheres != kNone && sc.lds_msphere_aux != kNone && sc.lds_spheres_tab != kNone);
            __builtin_assume(sc.lds_sphere_aux != kNone && sc.lds_materials != kNone);
        }
#endif
    } else {
        sc.lds_mspheres = sc.lds_msphere_aux = sc.lds_sphere_aux = kNone;
        if constexpr (T::BLOCK < RT_BIG_BLOCK) sc.lds_spheres_tab = kNone;
        if constexpr (!T::RICH) sc.lds_perlin = kNone;
#ifndef RT_NO_ASSUME
        if constexpr (T::BATCH && T::BLOCK >= RT_BIG_BLOCK) {  // the deep kernel is launched only with all of these staged (launch_one)
            __builtin_assume(sc.lds_boxes != kNone && sc.lds_objects != kNone && sc.lds_xforms != kNone);
            __builtin_assume(sc.lds_media != kNone && sc.lds_materials != kNone && sc.lds_perlin != kNone);
            __builtin_assume(sc.lds_spheres_tab != kNone && sc.lds_group_boxes != kNone);
            if constexpr (T::SEG) __builtin_assume(sc.lds_fast_order != kNone && sc.lds_seg_media != kNone && sc.lds_seg_cand != kNone);
        }
#endif
    }
    // ---- chip-resident working set ----
    NodeView nv{};
    uint16_t *queue = nullptr;
    if constexpr (T::WORLD == 0) {
        // BVH nodes in LDS, one 72-byte row each (see lds_node_f64 for the layout and why 72)
        nv.global = sc.nodes;
        nv.in_lds = (T::BATCH && T::BLOCK >= RT_BIG_BLOCK) ? true : a.lds_nodes != 0;  // the deep kernel: always (launch_one)
        if constexpr (T::FAST) {
            {  // the library's own tree: rows copied as they are (always staged: see walk_node_fast)
                const uint32_t words = sc.n_fast_nodes * (kFastNodeBytes / 4u);
                uint32_t *dst = reinterpret_cast<uint32_t *>(lds_raw);
                const uint32_t *src = reinterpret_cast<const uint32_t *>(sc.fast_rows);
                for (uint32_t k = threadIdx.x; k < words; k += blockDim.x) dst[k] = src[k];
                nv.n = sc.n_fast_nodes;
                auto stage = [](uint32_t off, const void *table, uint32_t bytes) {
                    if (off == kNone) return;
                    uint32_t *d2 = reinterpret_cast<uint32_t *>(lds_raw + off);
                    const uint32_t *s2 = static_cast<const uint32_t *>(table);
                    for (uint32_t w = threadIdx.x; w < bytes / 4u; w += blockDim.x) d2[w] = s2[w];
                };
                stage(sc.lds_mspheres, sc.mspheres, sc.n_mspheres * (uint32_t)sizeof(MSphereGeom));
                stage(sc.lds_msphere_aux, sc.msphere_aux, sc.n_mspheres * (uint32_t)sizeof(SphereAux));
                stage(sc.lds_spheres_tab, sc.spheres, sc.n_spheres * (uint32_t)sizeof(SphereGeom));
                stage(sc.lds_sphere_aux, sc.sphere_aux, sc.n_spheres * (uint32_t)sizeof(SphereAux));
                stage(sc.lds_materials, sc.materials, sc.n_materials * (uint32_t)sizeof(MaterialRec));
                __syncthreads();
            }
        } else if constexpr (T::SEG) {
            // segmented walk: the library's tree over the surface leaves (rows as they are), the leaf positions per node, the media
            auto copy = [](uint32_t off, const void *table, uint32_t bytes) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(lds_raw + off);
                const uint32_t *src = static_cast<const uint32_t *>(table);
                for (uint32_t w = threadIdx.x; w < bytes / 4u; w += blockDim.x) dst[w] = src[w];
            };
            copy(0u, sc.fast_rows, sc.n_fast_nodes * kFastNodeBytes);
            copy(sc.lds_fast_order, sc.fast_order, sc.n_fast_nodes * (uint32_t)sizeof(FastOrder));
            copy(sc.lds_seg_media, sc.seg_media, sc.n_seg_media * (uint32_t)sizeof(SegMedium));
            copy(sc.lds_seg_cand, sc.seg_cand, sc.n_seg_cand * (uint32_t)sizeof(SegCandidate));
            nv.n = sc.n_fast_nodes;
        } else if (nv.in_lds) {
            const uint32_t n = sc.n_world_nodes;
            for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
                BvhNodeRec node = sc.nodes[k];
                double *row = reinterpret_cast<double *>(lds_raw + k * kLdsNodeBytes);
                uint32_t *words = reinterpret_cast<uint32_t *>(lds_raw + k * kLdsNodeBytes + 48u);
                row[0] = node.xlo; row[1] = node.xhi; row[2] = node.ylo; row[3] = node.yhi; row[4] = node.zlo; row[5] = node.zhi;
                words[0] = node.a; words[1] = node.b; words[2] = node.escape;
            }
            nv.n = n;
            __syncthreads();
        }
    }
    if constexpr (T::COMPOSITE) {
        // small scenes: the tables of the composite leaf test, word by word, behind the node rows (see DeviceScene)
        auto stage = [](uint32_t off, const void *table, uint32_t bytes) {
            if (off == kNone) return;
            uint32_t *dst = reinterpret_cast<uint32_t *>(lds_raw + off);
            const uint32_t *src = static_cast<const uint32_t *>(table);
            for (uint32_t w = threadIdx.x; w < bytes / 4u; w += blockDim.x) dst[w] = src[w];
        };
        stage(sc.lds_quad_aa, sc.quad_aa, sc.n_quads * (uint32_t)sizeof(AAQuad));
        stage(sc.lds_boxes, sc.boxes, sc.n_boxes * (uint32_t)sizeof(BoxRec));
        stage(sc.lds_objects, sc.objects, sc.n_objects * (uint32_t)sizeof(ObjectRec));
        stage(sc.lds_xforms, sc.xforms, sc.n_xforms * (uint32_t)sizeof(Xform));
        stage(sc.lds_media, sc.media, sc.n_media * (uint32_t)sizeof(MediumRec));
        stage(sc.lds_spheres_tab, sc.spheres, sc.n_spheres * (uint32_t)sizeof(SphereGeom));
        stage(sc.lds_group_boxes, sc.group_boxes, sc.n_group_boxes * (uint32_t)sizeof(GroupBox));
        stage(sc.lds_materials, sc.materials, sc.n_materials * (uint32_t)sizeof(MaterialRec));
        if constexpr (T::RICH) stage(sc.lds_perlin, sc.perlin, sc.n_perlin * (uint32_t)sizeof(PerlinRec));
        __syncthreads();  // uniform: every thread of the block gets here
    }
    SphereView sv{};
    if constexpr (T::WORLD == 2) {
        queue = reinterpret_cast<uint16_t *>(lds_raw) + (threadIdx.x >> 6) * (kQueueCap * 64);
        sv.global = sc.spheres;
        sv.n = sc.n_spheres;
        sv.n_padded = (sc.n_spheres + 63u) & ~63u;
        sv.in_lds = a.lds_spheres != 0;
        sv.reach = sc.scan_reach;
        sv.exact = a.exact_scan != 0;
        if (sv.in_lds) {
            sv.planes_off = ((uint32_t)T::BLOCK / 64u) * kQueueCap * 64u * (uint32_t)sizeof(uint16_t);
            double *planes = reinterpret_cast<double *>(lds_raw + sv.planes_off);
            const uint32_t np = sv.n_padded;
            for (uint32_t k = threadIdx.x; k < np; k += blockDim.x) {
                SphereGeom g = sc.spheres[k < sv.n ? k : 0];
                planes[k] = g.cx; planes[np + k] = g.cy; planes[2 * np + k] = g.cz; planes[3 * np + k] = g.r2;
                planes[4 * np + k] = sc.sphere_scan[k < sv.n ? k : 0].k;
            }
            __syncthreads();
        }
    }

    // ---- persistent waves: every lane pulls pixels from one atomic cursor until the frame is done ----
    // Slots are numbered tile-major (8x8 tiles, row-major inside a tile) so that lanes refilled together
    // start on neighbouring pixels; a slot outside the frame is simply skipped.
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tiles_x = ((uint32_t)a.width + 7u) >> 3;
    // the work queue: tile-major slots of this rank's rows, or the entries of a pixel list (see RenderArgs::pixel_list)
    const uint32_t total_slots = a.pixel_list ? *(const RT_CONST uint32_t *)(uintptr_t)a.pixel_list_count
                                              : tiles_x * (((uint32_t)a.rows_owned + 7u) >> 3) * 64u;
    if (a.wave_priority == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.wave_priority == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.wave_priority == 3) __builtin_amdgcn_s_setprio(3);
    const CameraRec *__restrict__ cam = sc.camera;

    // this wave's role (wave-uniform): serve the list of heavy pixels first, a few at a time (RenderArgs::heavy_list)
    bool heavy_mode = T::ROLES && a.heavy_list != nullptr && (int)(threadIdx.x >> 6) < a.heavy_waves;
    bool heavy_dry = false;
    const uint32_t heavy_total = heavy_mode ? *(const RT_CONST uint32_t *)(uintptr_t)a.heavy_count : 0u;
    const uint32_t super_total = (heavy_mode && a.super_list) ? *(const RT_CONST uint32_t *)(uintptr_t)a.super_count : 0u;
    bool solo_phase = heavy_mode && a.super_list != nullptr;  // this wave still looks at the list of the longest chains first (RenderArgs::super_list)
    bool solo_hold = false;  // ... and holds one of them: no other pixel joins it
    // Pixels per serving wave: no more than it takes to start every listed pixel at once (a rank's stripes of a split frame list
    // few, and a chain is shortest with the wave to itself), at most what the launcher allows; the grouped scan of sphere lists
    // deals lanes in powers of two.
    int super_ppw = a.super_ppw, heavy_ppw = a.heavy_ppw;
    if (heavy_mode && a.adaptive_ppw) {
        const uint32_t serving = gridDim.x * (uint32_t)a.heavy_waves;
        auto fit = [&](uint32_t total, int cap) {
            int p = (int)((total + serving - 1u) / serving);
            p = p < 1 ? 1 : p;
            if (T::WORLD == 2) {
                int q = 1;
                while (q < p) q *= 2;
                p = q;
            }
            return p > cap ? cap : p;
        };
        // (both tiers by the two lists' sum: with the first tier alone in view a full C2 frame would serve it two to a wave and
        // start the second tier later -- 193 -> 197 ms)
        super_ppw = fit(super_total + heavy_total, a.super_ppw);
        heavy_ppw = fit(super_total + heavy_total, a.heavy_ppw);
    }
    if (heavy_mode) {
        if (a.heavy_priority == 1) __builtin_amdgcn_s_setprio(1);
        else if (a.heavy_priority == 2) __builtin_amdgcn_s_setprio(2);
        else if (a.heavy_priority == 3) __builtin_amdgcn_s_setprio(3);
    }
    bool active = false, exhausted = false;
    int i = 0, j = 0;
    size_t local = 0;
    Xorwow rng{0, 0, 0, 0, 0, 0};
    Vec col = mk(0.0, 0.0, 0.0);
    Vec throughput = mk(1.0, 1.0, 1.0), accumulated = mk(0.0, 0.0, 0.0);
    Ray ray{};
    int sample = 0, depth = 0;
    uint32_t wave_rays = 0;  // rays of the whole wave (uniform)

    uint32_t pix_rays = 0;  // rays this lane's current pixel has traced so far
    uint32_t my_tile = 0;   // tile of the current pixel (cost probe)
    int boost_left = 0;     // extra overdue-only passes still allowed before the next pixel-parallel pass
    // BVH worlds: per-lane resumable traversal (see Walk) and the hit it has found so far
    Walk walk{};
    walk.state = kNone;  // no walk in progress
    HitInfo walk_best;
    walk_best.t = 0.0;
    walk_best.ref = kNone;
    walk_best.obj = kNone;
    [[maybe_unused]] SegState seg{0u, kSegEnd, 0u};  // segmented walk: hi == kSegEnd also means "nothing more to walk" for an idle lane

#if RT_STAMP
    if (threadIdx.x == 0 && blockIdx.x == 0 && !a.probe) atomicMin(a.ray_counter + 5, (unsigned long long)wall_clock64());
#endif
#if RT_PHASES
    PhaseSums ph{};
    const unsigned long long ph_start = __builtin_readcyclecounter();
    bool ph_serving = heavy_mode;  // RT_PHASES == 2: only the waves serving heavy pixels report, and only that part of their life
#endif
    for (;;) {
        // A pixel that has used up its ray budget is "overdue": its samples cannot be spread over lanes (one
        // sequential RNG stream per pixel), and at one ray per pixel-parallel pass it would finish long after the
        // rest of the frame.  So after every pixel-parallel pass the wave inserts up to `boost_rounds` extra passes
        // in which only the overdue lanes advance, each of their rays scanned cooperatively by all 64 lanes.
        unsigned long long overdue = 0;
        bool boost = false;
        if constexpr (T::WORLD == 2) {
            overdue = __ballot(active && pix_rays >= a.ray_budget);
            if (a.overdue_priority) {  // diagnostic alternative: keep the pixel-parallel scan but raise this wave's issue priority
                if (overdue) __builtin_amdgcn_s_setprio(3);
                else __builtin_amdgcn_s_setprio(0);
                overdue = 0;
            }
            boost = overdue != 0 && boost_left > 0;
            if (boost) boost_left--;
            else boost_left = a.boost_rounds;
        }

        const bool from_list = heavy_mode && !heavy_dry;  // this refill takes heavy pixels off the list
        if (solo_hold && !__any(active)) solo_hold = false;
        if (!exhausted && !boost && (!heavy_mode || from_list) && !solo_hold) {
            unsigned long long need = __ballot(!active);
            // pixels_per_wave < 64: only the first lanes take pixels.  Sphere-list kernel: the others lend themselves to the
            // grouped scan; BVH kernels: the few rays have the wave's phases to themselves (shorter chain per pixel).
            const bool from_super = from_list && solo_phase;  // one of the longest chains, alone in this wave until it is done
            const int ppw = from_super ? super_ppw : (from_list ? heavy_ppw : a.pixels_per_wave);
            if (ppw < 64) need &= (1ull << ppw) - 1ull;
            if (need) {
                PH_BEGIN();
                [[maybe_unused]] const bool ph_was_idle = !active;
                const uint32_t cnt = (uint32_t)__popcll(need);
                const uint32_t queue_len = from_super ? super_total : (from_list ? heavy_total : total_slots);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(from_super ? a.super_cursor : (from_list ? a.heavy_cursor : a.cursor), cnt);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base + cnt >= queue_len) {
                    if (from_super) solo_phase = false;
                    else if (from_list) heavy_dry = true;
                    else exhausted = true;
                }
#if RT_STAMP
                if (exhausted && lane == 0 && !a.probe) atomicMin(a.ray_counter + 2, (unsigned long long)wall_clock64());
#endif
                const uint32_t slot = base + (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                if (((need >> lane) & 1ull) && slot < queue_len) {
                    // heaviest tiles first when the launcher has ranked them (see rt_render_launch); else row-major
                    const uint32_t w = slot & 63u;
                    uint32_t tile = (RT_ORDER_ON && a.tile_order && !a.pixel_list && !from_list) ? a.tile_order[slot >> 6] : slot >> 6;
                    int pi = (int)((tile % tiles_x) * 8u + (w & 7u));
                    int lr = (int)((tile / tiles_x) * 8u + (w >> 3));
                    bool take = pi < a.width && lr < a.rows_owned;
                    if (a.pixel_list || from_list) {  // listed pixels: compact index -> row, column
                        const uint32_t loc = from_super ? a.super_list[slot] : (from_list ? a.heavy_list[slot] : a.pixel_list[slot]);
                        lr = (int)(loc / (uint32_t)a.width);
                        pi = (int)(loc % (uint32_t)a.width);
                        tile = ((uint32_t)lr >> 3) * tiles_x + ((uint32_t)pi >> 3);
                        take = true;
                    } else if (a.pix_class && take) {
                        take = a.pix_class[(size_t)lr * (size_t)a.width + (size_t)pi] == 0;  // the other launch's pixel
                    }
                    if (take) {
                        i = pi;
                        j = owned_row(lr, a.stripe_rows, a.rank, a.world_size);
                        local = (size_t)lr * (size_t)a.width + (size_t)pi;
                        rng.d = a.state[0 * (size_t)a.n_pixels + local];
                        rng.v0 = a.state[1 * (size_t)a.n_pixels + local];
                        rng.v1 = a.state[2 * (size_t)a.n_pixels + local];
                        rng.v2 = a.state[3 * (size_t)a.n_pixels + local];
                        rng.v3 = a.state[4 * (size_t)a.n_pixels + local];
                        rng.v4 = a.state[5 * (size_t)a.n_pixels + local];
                        col = mk(0.0, 0.0, 0.0);
                        if (a.accum && a.spp_before > 0)  // progressive: continue the running sum in the same order
                            col = mk(a.accum[local * 3 + 0], a.accum[local * 3 + 1], a.accum[local * 3 + 2]);
                        throughput = mk(1.0, 1.0, 1.0);
                        accumulated = mk(0.0, 0.0, 0.0);
                        sample = 0;
                        depth = 0;
                        pix_rays = 0;
                        if (RT_PROBE_ON) my_tile = tile;
                        ray = camera_ray(cam, i, j, a.width, a.height, rng);
                        active = true;
#if RT_STAMP
                        if (a.dbg_times && !a.probe) a.dbg_times[2 * local] = (uint32_t)wall_clock64();
#endif
                        if constexpr (T::PARK) {
                            park_put_int<T::BLOCK>(sc.lds_park, 0, (uint32_t)i);
                            park_put_int<T::BLOCK>(sc.lds_park, 1, (uint32_t)j);
                            park_put_int<T::BLOCK>(sc.lds_park, 2, (uint32_t)local);
                            park_put_int<T::BLOCK>(sc.lds_park, 3, 0u);
                            park_put_int<T::BLOCK>(sc.lds_park, 4, 0u);
                            park_put_vec<T::BLOCK>(sc.lds_park, 0, col);
                            park_put_vec<T::BLOCK>(sc.lds_park, 3, throughput);
                            park_put_vec<T::BLOCK>(sc.lds_park, 6, accumulated);
                            park_put_rng<T::BLOCK>(sc.lds_park, rng);
                        }
                        if constexpr (T::WORLD == 0) {
                            walk_begin<T::FAST || T::SEG>(walk, ray, DBL_MAX);
                            if constexpr (T::SEG) {  // "between walks", before the first one: seg_advance at the head of the next round
                                seg = SegState{0u, 0u, 0u};
                                walk.state = kNone;
                            }
                        }
                    }
                }
                if (from_super && __any(active)) solo_hold = true;
                PH_END(3, ph_was_idle && active);
            }
        }
        const unsigned long long live = __ballot(active);
        if (!live) {
            if (heavy_mode && heavy_dry) {  // the list is done and so are this wave's heavy pixels: join the tile queue
                heavy_mode = false;
                __builtin_amdgcn_s_setprio(0);
#if RT_PHASES == 2
                ph_flush(a, ph, ph_start, lane);
                ph_serving = false;
#endif
#if RT_STAMP
                if (lane == 0 && !a.probe) atomicMax(a.ray_counter + 8, (unsigned long long)wall_clock64());  // last serving wave done with the heavy list
#endif
                continue;
            }
            if (exhausted) break;
            continue;
        }

        // ---- one ray segment for every lane in `todo` ----
        unsigned long long todo = live;
        HitInfo h;
        h.t = 0.0;
        h.ref = kNone;
        h.obj = kNone;
        bool hit = false;
        if constexpr (T::WORLD == 2) {
            if (boost) todo = overdue;
            if (!boost && a.pixels_per_wave >= 64 && __popcll(live) >= a.coop_threshold) {
                PH_BEGIN();
                if (active) hit = a.exact_scan    ? scan_uniform(sc, queue, lane, ray, 0.001, DBL_MAX, h)
                                 : a.filter_fp64 ? (sv.in_lds ? scan_filtered<true>(sc, sv.planes_off, sv.n_padded, queue, lane, ray, 0.001, DBL_MAX, h)
                                                              : scan_filtered<false>(sc, 0u, 0u, queue, lane, ray, 0.001, DBL_MAX, h))
                                 : sv.in_lds     ? scan_filtered32<true>(sc, sv.planes_off, sv.n_padded, queue, lane, ray, 0.001, DBL_MAX, h)
                                                 : scan_filtered32<false>(sc, 0u, 0u, queue, lane, ray, 0.001, DBL_MAX, h);
                PH_END(0, active);
            } else {
                PH_BEGIN();
                if (sv.in_lds && !a.coop_single) scan_grouped(sv, lane, todo, ray, 0.001, DBL_MAX, h, hit);
                else scan_cooperative(sv, lane, todo, ray, 0.001, DBL_MAX, h, hit);
                PH_END(1, (todo >> lane) & 1ull);
            }
        }
        bool thin = false;
        if constexpr (T::WORLD == 0 && !T::COMPOSITE) {
            // Frame tail of a sphere world: the queue is dry and few lanes are left -- scan instead of walking (scan_grouped_ms).
            thin = sc.ms_planes != nullptr && ((exhausted && __popcll(live) < a.coop_threshold) || (heavy_mode && a.heavy_scan));
            if (thin) {
                PH_BEGIN();
                scan_grouped_ms(sc, lane, live, ray, 0.001, DBL_MAX, h, hit);
                walk.state = kNone;  // a walk in progress is dropped: the scan has covered every leaf
                PH_END(1, active);
            }
        }
        if constexpr (T::WORLD == 0) if (!thin) {
            // Traversal burst: every walking lane advances up to kBurst nodes.  Lanes whose walk is complete
            // wait for shading; they are shaded once enough of them have gathered (or nobody is walking any
            // more), then start their next ray and rejoin the walkers.
            // Inner nodes and leaves run in separate phases so that the (long, branchy) leaf tests execute with
            // many lanes at once instead of trailing every node visit with a few.
            // node/leaf phase pairs per look at the shading queue: measured optimum 6 for primitive worlds (C3: 8 -> 6 is
            // +7 %, 4 is -1 %), 4 for composite ones (C5: +6 %; C4 indifferent)
            const int n_rounds = T::BATCH ? a.rounds : (T::COMPOSITE ? kRoundsComposite : (T::FAST ? kRoundsFast : kRounds));
            for (int round = 0; round < n_rounds; round++) {
                if constexpr (T::SEG) {
                    // between two walks (a new ray, or a walk that was not the ray's last has ended): the media that are due, then
                    // the next walk (seg_advance) -- the one place where media are tested and draw
                    const bool between = active && walk.state == kNone && seg.hi != kSegEnd;
                    if (__any(between)) {
                        PH_BEGIN();
#if RT_PHASES
                        const bool ph_again = between && seg.hi != 0u;  // not the ray's first time here: a limited walk has ended
#endif
                        if (between) seg_advance<T>(sc, ray, walk, walk_best, rng, seg PH_PASS);
                        PH_END(17, between);
#if RT_PHASES
                        if (__any(ph_again)) PH_END(12, ph_again);
#endif
                    }
                }
                for (int step = 0; step < (T::COMPOSITE ? a.node_burst : kBurst); step++) {
                    const bool mover = walk_moving(walk.state);
                    [[maybe_unused]] bool limited_walks = false;
                    if constexpr (T::SEG) limited_walks = __any(mover && seg.hi != kSegEnd);
#if RT_SIMPLE_BREAK
                    if (!__any(mover)) break;
#else
                    const int movers = __popcll(__ballot(mover));
                    const int parked = __popcll(__ballot(walk_parked(walk.state)));
                    // most walkers are waiting at leaves: go test them.  A composite leaf (box, instance, medium) costs
                    // tens of node steps, so there the leaf phase waits for a larger share of the walkers.
                    if (movers == 0 || movers * (T::COMPOSITE ? a.park_ratio : 1) < parked) break;
#endif
                    {
                        PH_BEGIN();
#if RT_PHASES
                        if (limited_walks) PH_END(13, mover && seg.hi != kSegEnd);
#endif
                        if (mover) {
                            if constexpr (T::FAST) {
                                walk_node_fast(ray, 0.001, walk);
                                PH_COUNT(14);
                                if (walk_moving(walk.state)) {
                                    walk_node_fast(ray, 0.001, walk);
                                    PH_COUNT(14);
                                }
                            } else if constexpr (T::SEG) {
                                if (limited_walks) {  // some lane of the wave is on its way to a medium (wave-uniform, rare)
                                    walk_node_seg(sc, ray, 0.001, walk, seg.hi);
                                    PH_COUNT(14);
                                    if (walk_moving(walk.state)) {
                                        walk_node_seg(sc, ray, 0.001, walk, seg.hi);
                                        PH_COUNT(14);
                                    }
                                } else {
                                    walk_node_open(ray, 0.001, walk);
                                    PH_COUNT(14);
                                    if (walk_moving(walk.state)) {
                                        walk_node_open(ray, 0.001, walk);
                                        PH_COUNT(14);
                                    }
                                }
                            } else
                            {
                                walk_node<T::BATCH>(nv, ray, 0.001, walk);
                                PH_COUNT(14);
                            }
                            // Primitive worlds (deep BVH, cheap leaves): a second visit before the next look at the
                            // wave's state -- the ballots and counts that steer the phases cost a fifth of a node visit.
                            // Not for composite worlds: the Cornell box's tree is three levels deep (measured -17 %).
                            // Kind-batched kernels run on deep trees too: the same second visit (C5 +x %).
                            if constexpr ((!T::COMPOSITE || T::BATCH) && !T::FAST && !T::SEG) {
                                if (walk_moving(walk.state)) {
                                    walk_node<T::BATCH>(nv, ray, 0.001, walk);
                                    PH_COUNT(14);
                                }
                            }
                        }
                        PH_END(0, mover);
                    }
                }
                if constexpr (T::BATCH) {
                    const bool at_leaf = walk_parked(walk.state);
                    if (__any(at_leaf)) {
                        const uint32_t kind = (walk.state >> kWalkKindShift) & 3u;  // of the pending leaf (walk_node, walk_leaf_pass)
                        // everything is served in the last round before shading is looked at again, and when nobody walks
                        const bool serve_all = round + 1 == n_rounds || !__any(walk_moving(walk.state));
#define RT_LEAF_PASS(K, SLOT)                                                                                          \
                        {                                                                                              \
                            const bool mine = at_leaf && kind == (K);                                                  \
                            const int waiting = __popcll(__ballot(mine));                                              \
                            if (waiting > 0 && (serve_all || waiting >= a.leaf_batch)) {                               \
                                PH_BEGIN();                                                                            \
                                if (mine) walk_leaf_pass<T, (K)>(sc, nv, ray, 0.001, walk, walk_best, rng,             \
                                                                 T::SEG ? seg_pending_leaf(walk.state) : walk_pending_leaf(nv, walk.state) PH_PASS, \
                                                                 false, false, 0.0, kNone, seg.lo, seg.hi);            \
                                PH_END(SLOT, mine);                                                                    \
                            }                                                                                          \
                        }
                        RT_LEAF_PASS(LK_BOX, 16)
                        if constexpr (T::MEDIA && !T::SEG) RT_LEAF_PASS(LK_MEDIUM, 17)  // segmented walk: no medium is a leaf of the tree
                        {  // instances and groups: the whole wave takes part (walk_object_pass)
                            const bool mine = at_leaf && kind == LK_OBJECT;
                            const int waiting = __popcll(__ballot(mine));
                            if (waiting > 0 && (serve_all || waiting >= a.object_batch)) {
                                PH_BEGIN();
                                walk_object_pass<T>(sc, nv, ray, 0.001, walk, walk_best, rng,
                                                    mine ? (T::SEG ? seg_pending_leaf(walk.state) : walk_pending_leaf(nv, walk.state)) : kNone, mine,
                                                    lane PH_PASS, seg.lo, seg.hi);
                                PH_END(18, mine);
                            }
                        }
                        RT_LEAF_PASS(LK_PRIM, 19)
#undef RT_LEAF_PASS
                    }
                } else {
                    const bool at_leaf = walk_parked(walk.state);
                    PH_BEGIN();
                    if (at_leaf) {
                        if constexpr (T::FAST) walk_leaves_fast(sc, ray, 0.001, walk, walk_best);
                        else walk_leaves<T>(sc, nv, ray, 0.001, walk, walk_best, rng PH_PASS);
                    }
                    if (__any(at_leaf)) PH_END(1, at_leaf);
                }
                if constexpr (T::SEG) {
                    if (!__any(walk.state != kNone || (active && seg.hi != kSegEnd))) break;
                } else
                if (!__any(walk.state != kNone)) break;
            }
            // segmented walk: a lane between two walks is still on its way, not waiting to be shaded
            const unsigned long long walkers = T::SEG ? __ballot(walk.state != kNone || (active && seg.hi != kSegEnd)) : __ballot(walk.state != kNone);
            const unsigned long long waiting = live & ~walkers;
            const int n_wait = __popcll(waiting), n_walk = __popcll(walkers);
            todo = (n_wait >= a.shade_batch || n_wait >= n_walk) ? waiting : 0ull;
            hit = walk.any;
            h = walk_best;
        }
#if RT_PHASES
        const unsigned long long ph_ts = __builtin_readcyclecounter();
#endif
        if constexpr (T::GROUPED) {  // every ray's leaves are dealt to 64 / pixels_per_wave lanes (the launcher keeps pixels_per_wave a power of two below 64)
            PH_BEGIN();
            scan_leaves_grouped<T>(sc, lane, 6 - (__ffs(a.pixels_per_wave) - 1), todo, ray, 0.001, DBL_MAX, h, hit, rng PH_PASS);
            PH_END(1, (todo >> lane) & 1ull);
        }
        if (a.max_depth > 0) wave_rays += (uint32_t)__popcll(todo);
        if ((todo >> lane) & 1ull) {
            const bool no_bounces = a.max_depth <= 0;  // R/kernel.cu:71: the bounce loop never runs, RayColor returns black
            if (!no_bounces) {
                if constexpr (T::PARK) {
                    if (RT_PROBE_ON && a.probe) park_put_int<T::BLOCK>(sc.lds_park, 4, park_get_int<T::BLOCK>(sc.lds_park, 4) + 1u);  // only the rehearsal asks
                } else {
                    pix_rays++;
                }
            }
            if constexpr (T::WORLD == 1 && !T::GROUPED) hit = world_hit_list<T>(sc, ray, 0.001, DBL_MAX, h, rng PH_PASS);
            if constexpr (T::PARK) {  // back from LDS: what the shading works on (no leaf of these worlds draws random numbers)
                throughput = park_get_vec<T::BLOCK>(sc.lds_park, 3);
                accumulated = park_get_vec<T::BLOCK>(sc.lds_park, 6);
                park_get_rng<T::BLOCK>(sc.lds_park, rng);
            }
            bool path_ends;
            if (no_bounces) {
                path_ends = true;
            } else if (!hit) {  // R/kernel.cu:74-79
                accumulated = accumulated + throughput * load3c(((const RT_CONST CameraRec *)(uintptr_t)cam)->bg);
                path_ends = true;
            } else {
#if RT_PHASES
                const unsigned long long ph_a = __builtin_readcyclecounter();
#endif
                Surface s = make_surface<T>(sc, ray, h);
#if RT_PHASES
                asm volatile("" ::"v"(s.p.x), "v"(s.n.z), "v"(s.mat));
                const unsigned long long ph_b = __builtin_readcyclecounter();
                ph.t[20] += ph_b - ph_a;
                ph.l[20] += (unsigned long long)__popcll(__ballot(true));
                ph.n[20] += 1ull;
#endif
                path_ends = !shade<T>(sc, s, ray, throughput, accumulated, rng);
#if RT_PHASES
                asm volatile("" ::"v"(ray.d.x), "v"(throughput.x));
                ph.t[21] += __builtin_readcyclecounter() - ph_b;
                ph.l[21] += (unsigned long long)__popcll(__ballot(true));
                ph.n[21] += 1ull;
#endif
                if (!path_ends && ++depth >= a.max_depth) path_ends = true;  // R/kernel.cu:71,97
            }
#if RT_PHASES
            const unsigned long long ph_c = __builtin_readcyclecounter();
            if (path_ends && sample + 1 < a.spp) {
                ph.l[22] += (unsigned long long)__popcll(__ballot(true));
                ph.n[22] += 1ull;
            }
#endif
            if (path_ends) {  // R/kernel.cu:143: col += RayColor(...)
                if constexpr (T::PARK) {
                    col = park_get_vec<T::BLOCK>(sc.lds_park, 0);
                    sample = (int)park_get_int<T::BLOCK>(sc.lds_park, 3);
                }
                col = col + accumulated;
                if (++sample < a.spp) {
                    if constexpr (T::PARK) {
                        park_put_vec<T::BLOCK>(sc.lds_park, 0, col);
                        park_put_int<T::BLOCK>(sc.lds_park, 3, (uint32_t)sample);
                        i = (int)park_get_int<T::BLOCK>(sc.lds_park, 0);
                        j = (int)park_get_int<T::BLOCK>(sc.lds_park, 1);
                    }
                    ray = camera_ray(cam, i, j, a.width, a.height, rng);
                    throughput = mk(1.0, 1.0, 1.0);
                    accumulated = mk(0.0, 0.0, 0.0);
                    depth = 0;
                } else if (RT_PROBE_ON && a.probe) {
                    // cost probe: the samples were a rehearsal (the saved RNG state is untouched); book the rays
                    if constexpr (T::PARK) {
                        local = (size_t)park_get_int<T::BLOCK>(sc.lds_park, 2);
                        pix_rays = park_get_int<T::BLOCK>(sc.lds_park, 4);
                        const uint32_t row = (uint32_t)local / (uint32_t)a.width, column = (uint32_t)local % (uint32_t)a.width;
                        my_tile = (row >> 3) * tiles_x + (column >> 3);
                    }
                    if (a.tile_cost) atomicAdd(a.tile_cost + my_tile, pix_rays);
                    if (a.pix_cost) a.pix_cost[local] = pix_rays;
                    active = false;
                } else {
                    // R/kernel.cu:146-153: save the RNG state, average, gamma 2
#if RT_STAMP
                    if (a.dbg_times) a.dbg_times[2 * local + 1] = (uint32_t)wall_clock64();
#endif
                    if constexpr (T::PARK) local = (size_t)park_get_int<T::BLOCK>(sc.lds_park, 2);
                    a.state[0 * (size_t)a.n_pixels + local] = rng.d;
                    a.state[1 * (size_t)a.n_pixels + local] = rng.v0;
                    a.state[2 * (size_t)a.n_pixels + local] = rng.v1;
                    a.state[3 * (size_t)a.n_pixels + local] = rng.v2;
                    a.state[4 * (size_t)a.n_pixels + local] = rng.v3;
                    a.state[5 * (size_t)a.n_pixels + local] = rng.v4;
                    if (a.accum) {
                        a.accum[local * 3 + 0] = col.x;
                        a.accum[local * 3 + 1] = col.y;
                        a.accum[local * 3 + 2] = col.z;
                    }
                    col = over(col, (double)(a.spp_before + a.spp));
                    a.pixels[local * 3 + 0] = sqrt(col.x);
                    a.pixels[local * 3 + 1] = sqrt(col.y);
                    a.pixels[local * 3 + 2] = sqrt(col.z);
                    active = false;
                }
            }
#if RT_PHASES
            asm volatile("" ::"v"(ray.d.x), "v"(col.x));
            ph.t[22] += __builtin_readcyclecounter() - ph_c;  // next camera ray or pixel done (booked together)
#endif
            if constexpr (T::PARK) {
                if (active) {
                    park_put_vec<T::BLOCK>(sc.lds_park, 3, throughput);
                    park_put_vec<T::BLOCK>(sc.lds_park, 6, accumulated);
                    park_put_rng<T::BLOCK>(sc.lds_park, rng);
                }
            }
            if constexpr (T::WORLD == 0) {
                if (active) {
                    walk_begin<T::FAST || T::SEG>(walk, ray, DBL_MAX);
                    if constexpr (T::SEG) {
                        seg = SegState{0u, 0u, 0u};
                        walk.state = kNone;
                    }
                } else if constexpr (T::SEG) {
                    seg.hi = kSegEnd;  // no pixel: nothing between walks either
                }
            }
        }
#if RT_PHASES
        if (todo) {
            ph.t[2] += __builtin_readcyclecounter() - ph_ts;
            ph.l[2] += (unsigned long long)__popcll(todo);
            ph.n[2] += 1ull;
        }
#endif
    }
#if RT_PHASES
    if (RT_PHASES == 1 || ph_serving) ph_flush(a, ph, ph_start, lane);
#endif

#if RT_STAMP
    if (lane == 0 && !a.probe) {
        atomicMax(a.ray_counter + 3, (unsigned long long)wall_clock64());
        atomicMin(a.ray_counter + 4, (unsigned long long)wall_clock64());  // first wave to finish
    }
#endif
    // one atomic per wave for the ray counter
    const unsigned long long total = wave_rays;
    if (lane == 0 && total) atomicAdd(a.ray_counter, total);
}

#if RT_STRICT && RT_GROUP == 0
// Rank the tiles by probed cost, heaviest first.  A pixel's samples are sequential (one RNG stream), so the frame can
// never end before its longest pixel does: those pixels have to start first, not wherever row-major order puts them.
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t *cost, uint32_t *order, uint32_t n, uint32_t flat_x8)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t peak;
    __shared__ unsigned long long total;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        peak = 1;
        total = 0;
    }
    __syncthreads();
    uint32_t mx = 0;
    unsigned long long sum = 0;
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
        mx = cost[k] > mx ? cost[k] : mx;
        sum += cost[k];
    }
    atomicMax(&peak, mx);
    atomicAdd(&total, sum);
    __syncthreads();
    const uint32_t top = peak;
    // Nothing to gain where no tile stands out: keep the row-major order.  flat_x8 / 8 = how far the heaviest tile must be above
    // the mean for the order to matter (launcher's choice: 4 where only the longest chains count, less where the spread of
    // ordinary tiles decides the last generation of pixels, see rt_render_launch).
    if ((unsigned long long)top * n * 8ull <= (unsigned long long)flat_x8 * total) {
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) order[k] = k;
        return;
    }
    auto klass = [top](uint32_t c) { return 255u - (uint32_t)(((unsigned long long)c * 255ull) / top); };  // 0 = heaviest
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) atomicAdd(&hist[klass(cost[k])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 256; b++) {
            uint32_t c = hist[b];
            hist[b] = run;
            run += c;
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) order[atomicAdd(&hist[klass(cost[k])], 1u)] = k;
}

__global__ __launch_bounds__(256) void classify_pixels_kernel(const uint32_t *cost, uint32_t n, uint32_t threshold, uint8_t *klass,
                                                               uint32_t *list, uint32_t *count, uint32_t *super_list, uint32_t super_threshold,
                                                               uint32_t width, uint32_t near_percent, uint32_t near_neighbours)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    bool heavy = cost[k] >= threshold;
    // The rehearsal is a handful of samples of a heavy-tailed count: a pixel a little under the threshold whose neighbours are over
    // it is more likely a long chain that looked short than a short one (long chains come in patches -- a glass sphere) and goes on
    // the list with them.  A pixel wrongly left with the light ones runs at their pace to the frame's very end.
    if (!heavy && near_neighbours > 0 && width > 0 && cost[k] * 100u >= threshold * near_percent) {
        const uint32_t row = k / width, column = k % width, rows = n / width;
        uint32_t over = 0;
        for (int dr = -1; dr <= 1; dr++)
            for (int dc = -1; dc <= 1; dc++) {
                const int r = (int)row + dr, c = (int)column + dc;
                if ((dr || dc) && r >= 0 && c >= 0 && r < (int)rows && c < (int)width) over += cost[(uint32_t)r * width + (uint32_t)c] >= threshold ? 1u : 0u;
            }
        heavy = over >= near_neighbours;
    }
    const bool longest = heavy && super_list && cost[k] >= super_threshold;
    klass[k] = heavy ? 1 : 0;
    if (longest) super_list[atomicAdd(count + 1, 1u)] = k;
    else if (heavy) list[atomicAdd(count, 1u)] = k;  // order within the list is irrelevant: every listed pixel starts at once
}

hipError_t launch_classify_pixels(const uint32_t *pix_cost, uint32_t n_pixels, uint32_t threshold, uint8_t *pix_class, uint32_t *list,
                                  uint32_t *count, hipStream_t stream, uint32_t *super_list, uint32_t super_threshold, uint32_t width,
                                  uint32_t near_percent, uint32_t near_neighbours)
{
    if (n_pixels == 0) return hipSuccess;
    hipLaunchKernelGGL(classify_pixels_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0, stream, pix_cost, n_pixels, threshold,
                       pix_class, list, count, super_list, super_threshold, width, near_percent, near_neighbours);
    return hipGetLastError();
}

hipError_t launch_tile_order(const uint32_t *tile_cost, uint32_t *tile_order, uint32_t n_tiles, uint32_t flat_x8, hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, tile_cost, tile_order, n_tiles, flat_x8);
    return hipGetLastError();
}
#endif

#if RT_STRICT
#define RT_SUFFIX strict
#else
#define RT_SUFFIX fast
#endif
#define RT_CAT2(a, b) a##b
#define RT_CAT(a, b) RT_CAT2(a, b)

#if RT_GROUP == 0
hipError_t RT_CAT(launch_seed_, RT_SUFFIX)(const SeedArgs &a, hipStream_t stream)
{
    if (a.n_pixels == 0) return hipSuccess;
    dim3 grid((a.n_pixels + 255u) / 256u), block(256);
    hipLaunchKernelGGL(seed_kernel<RT_STRICT>, grid, block, 0, stream, a);
    return hipGetLastError();
}
#endif

namespace {
#ifndef RT_WAVES_SPHERES
#define RT_WAVES_SPHERES 3  // three workgroups per CU serve the heavy and the light pixels (rt_render_launch): 168 VGPRs at most
#endif
#ifndef RT_WAVES_BVH
#define RT_WAVES_BVH 3
#endif
using TSphereList = Traits<2, false, false, RT_WAVES_SPHERES>;
using TBvhPrims = Traits<0, false, false, RT_WAVES_BVH>;
#ifndef RT_BLOCK_FAST
#define RT_BLOCK_FAST 768
#endif
using TBvhPrimsFast = Traits<0, false, false, RT_WAVES_BVH, false, false, false, RT_BLOCK_FAST, true>;  // through the library's own tree
#ifndef RT_WAVES_GENERAL
#define RT_WAVES_GENERAL 2
#endif
#ifndef RT_WAVES_INSTANCES
#define RT_WAVES_INSTANCES 3  // 168 VGPRs: the instances kernel sits right at the step from three waves per SIMD to two
#endif
using TBvhGeneral = Traits<0, true, true, RT_WAVES_GENERAL>;
using TListGeneral = Traits<1, true, true, RT_WAVES_GENERAL>;
using TBvhInstances = Traits<0, true, false, RT_WAVES_INSTANCES, false>;  // instances / boxes, no media, plain textures (C4)
// The media and general kernels need ~220 and ~300 VGPRs.  Walking a deep tree they are latency-bound -- one wave per
// SIMD runs at half the speed of two -- so three waves with a hundred-odd registers spilled to scratch still win
// (Cornell smoke +5 %, C5 +4.5 %).  A shallow world with expensive shading (Perlin, image texture) loses a third that
// way, so the general kernel exists in both shapes and the launcher picks by the depth of the world's tree.
#ifndef RT_WAVES_MEDIA
#define RT_WAVES_MEDIA 3
#endif
using TBvhMedia = Traits<0, true, false, RT_WAVES_MEDIA, true>;                      // + ConstantMedium (Cornell smoke)
#ifndef RT_WAVES_DEEP
#define RT_WAVES_DEEP 3
#endif
#ifndef RT_BLOCK_DEEP
#define RT_BLOCK_DEEP 768
#endif
using TBvhGeneralDeep = Traits<0, true, true, RT_WAVES_DEEP, true, true, false, RT_BLOCK_DEEP>;
// ... and walked through the library's tree, one walk per run of surface leaves between two media (Traits::SEG): worlds that
// flat_scene.h SCENE_SEGMENTED describes (C5)
using TBvhSegmented = Traits<0, true, true, RT_WAVES_DEEP, true, true, false, RT_BLOCK_DEEP, true>;
// List scans over primitives / instances without media or table-walking textures.  Also the BVH worlds of small
// scenes: for up to 16 leaves within a cost budget (FlatScene::scan_cost) a scan of all of them in the tree's leaf order -- every lane on the same leaf, rows
// through uniform loads, no node visits, no phases -- beats walking the tree (Cornell box: 8 leaves, 7 nodes).  Without
// media no leaf draws random numbers, so the closest hit is the one the walk finds (the reference's own BVH = list
// invariant; `tests/test_parity_gpu.py::test_small_world_scan_equals_the_bvh_walk`).
// General nesting (REF_TREE leaves, tree_hit): the general kernels plus the interpreter.  Its stack of 16 frames lives in
// scratch -- the reference's own recursion needs a 32 KiB stack per thread (R/kernel.cu:599) -- so these instantiations
// are only ever launched for scenes that nest objects beyond what ObjectRec expresses (none of the ten built-in scenes).
using TBvhNested = Traits<0, true, true, 2, true, false, true>;
using TListNested = Traits<1, true, true, 2, true, false, true>;
using TListPrims = Traits<1, false, false, 4>;  // 127-129 VGPRs without the bound: the strict build would drop to three waves for one register
#ifndef RT_WAVES_LIST_INSTANCES
#define RT_WAVES_LIST_INSTANCES 4  // 128 VGPRs and 52 B of scratch for a fourth wave per SIMD: C4 +2.4 % (139 VGPRs, none, three waves before)
#endif
using TListInstances = Traits<1, true, false, RT_WAVES_LIST_INSTANCES, false>;
// The same kernel compiled for FIVE waves per SIMD (96 VGPRs and 64 B of scratch with the path state parked in LDS, Traits::PARK).  Its passes take 1.38 times as long
// -- the SIMDs' issue slots are nearly full with four waves -- so in the steady state it is the slower of the two; but a pixel's
// samples are one chain, pixels of these worlds all cost about the same, and a frame is therefore a whole number of pixel
// "generations" on the resident lanes: 800 x 800 pixels are 2.44 generations on the 262 144 lanes of four waves per SIMD -- three,
// the last 44 % full -- and 1.95 on the 327 680 of five: two.  dispatch() picks by that count (list_instances_waves).
using TListInstances5 = Traits<1, true, false, 5, false>;
// The same two with the leaves of every ray dealt to lanes (pixels_per_wave < 64: fewer pixels than lanes, the frame is bound
// by the latency of a ray, not by throughput -- registers matter more than a fourth wave)
#ifndef RT_WAVES_GROUPED
#define RT_WAVES_GROUPED 3
#endif
using TListPrimsGrouped = Traits<1, false, false, RT_WAVES_GROUPED, false, false, false, 256, false, true>;
using TListInstancesGrouped = Traits<1, true, false, RT_WAVES_GROUPED, false, false, false, 256, false, true>;

// Do the node rows and the sphere / material rows of a primitive world fit the library-tree kernel's LDS?  (The same sums as
// launch_one<TBvhPrimsFast>'s placement, for callers that have no reference tree to fall back to.)
[[maybe_unused]] static bool fast_rows_fit(const DeviceScene &sc)
{
    auto up = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t nodes = (size_t)sc.n_fast_nodes * kFastNodeBytes;
    if (nodes > 60 * 1024) return false;
    const size_t rows = up(nodes) + up((size_t)sc.n_mspheres * sizeof(MSphereGeom)) + up((size_t)sc.n_mspheres * sizeof(SphereAux)) +
                        up((size_t)sc.n_spheres * sizeof(SphereGeom)) + up((size_t)sc.n_spheres * sizeof(SphereAux)) +
                        up((size_t)sc.n_materials * sizeof(MaterialRec));
    return rows + 5 * 64 <= 158 * 1024;
}

template <class T>
hipError_t launch_one(const DeviceScene &sc_in, RenderArgs a, hipStream_t stream, KernelInfo *info)
{
    DeviceScene sc = sc_in;
    [[maybe_unused]] const RenderArgs a_in = a;
    if (!T::ROLES && a.heavy_list) return hipErrorInvalidValue;  // this instantiation has no serving waves (Traits::ROLES): its listed pixels would never be rendered
    auto kernel = render_kernel<RT_STRICT, T>;
    uint32_t tiles = (((uint32_t)a.width + 7u) >> 3) * (((uint32_t)a.rows_owned + 7u) >> 3);
    size_t lds = 0;
    a.lds_nodes = 0;
    if (T::WORLD == 0) {
        size_t need = (T::FAST || T::SEG) ? (size_t)sc.n_fast_nodes * kFastNodeBytes : (size_t)sc.n_world_nodes * kLdsNodeBytes;
        if (need <= 60 * 1024) {  // keep >= 2 workgroups (of 256 threads) per CU resident
            lds = need;
            a.lds_nodes = 1;
        }
        if (T::FAST && T::BLOCK >= RT_BIG_BLOCK && a.lds_nodes) {
            // One workgroup per CU: the sphere rows the leaf tests and the hit record read and the material rows follow the
            // node rows into the CU's LDS -- a frame ends with its longest pixel, and that pixel's chain is made of exactly
            // these dependent reads (C3: leaf pass 2100 -> ... cycles, shading pass 11000 -> ... cycles).
            const size_t budget = 158 * 1024;
            size_t off = (lds + 15) & ~(size_t)15;
            auto place = [&](uint32_t &slot, size_t bytes) {
                if (bytes == 0 || off + bytes + 64 > budget) return;
                slot = (uint32_t)off;
                off += (bytes + 15) & ~(size_t)15;
            };
            place(sc.lds_mspheres, (size_t)sc.n_mspheres * sizeof(MSphereGeom));
            place(sc.lds_msphere_aux, (size_t)sc.n_mspheres * sizeof(SphereAux));
            place(sc.lds_spheres_tab, (size_t)sc.n_spheres * sizeof(SphereGeom));
            place(sc.lds_sphere_aux, (size_t)sc.n_spheres * sizeof(SphereAux));
            place(sc.lds_materials, (size_t)sc.n_materials * sizeof(MaterialRec));
            lds = off;
        }
        if constexpr (T::FAST && T::BLOCK >= RT_BIG_BLOCK) {
            // The library-tree kernel reads these rows from LDS only (no global side in its accessors: head of
            // render_kernel); a world whose rows do not fit is walked by the reference-tree kernel.
            const bool fits = a.lds_nodes && (sc.n_mspheres == 0 || (sc.lds_mspheres != kNone && sc.lds_msphere_aux != kNone)) &&
                              (sc.n_spheres == 0 || (sc.lds_spheres_tab != kNone && sc.lds_sphere_aux != kNone)) &&
                              (sc.n_materials == 0 || sc.lds_materials != kNone);
            if (!fits) return launch_one<TBvhPrims>(sc_in, a_in, stream, info);
            auto empty = [](uint32_t &slot) { if (slot == kNone) slot = 0; };  // an empty table is never read
            empty(sc.lds_mspheres); empty(sc.lds_msphere_aux); empty(sc.lds_spheres_tab); empty(sc.lds_sphere_aux); empty(sc.lds_materials);
        }
        if constexpr (T::SEG) {  // the leaf positions per node and the media, right behind the node rows
            size_t off = (lds + 15) & ~(size_t)15;
            sc.lds_fast_order = (uint32_t)off;
            off += ((size_t)sc.n_fast_nodes * sizeof(FastOrder) + 15) & ~(size_t)15;
            sc.lds_seg_media = (uint32_t)off;
            off += ((size_t)(sc.n_seg_media ? sc.n_seg_media : 1u) * sizeof(SegMedium) + 15) & ~(size_t)15;
            sc.lds_seg_cand = (uint32_t)off;
            off += ((size_t)(sc.n_seg_cand ? sc.n_seg_cand : 1u) * sizeof(SegCandidate) + 15) & ~(size_t)15;
            lds = off;
        }
        if (T::COMPOSITE) {
            // Small tables ride along behind the node rows, each on its own merits: the records a leaf test or the shading
            // chases through (object -> transforms -> medium; material rows; Perlin tables: a few KB even in the Book-2
            // final scene) and, where they fit as well, the quad / box rows (Cornell box: 2 KB).
            // three 256-thread workgroups per CU share its 160 KB, or one of 768 threads has (nearly) all of it
            const size_t budget = T::BLOCK >= RT_BIG_BLOCK ? 158 * 1024 : 52 * 1024;
            size_t off = (lds + 15) & ~(size_t)15;
            auto place = [&](uint32_t &slot, size_t bytes, size_t cap) {
                if (bytes == 0 || bytes > cap || off + bytes + 64 > budget) return;
                slot = (uint32_t)off;
                off += (bytes + 15) & ~(size_t)15;
            };
            place(sc.lds_objects, (size_t)sc.n_objects * sizeof(ObjectRec), 4096);
            place(sc.lds_xforms, (size_t)sc.n_xforms * sizeof(Xform), 4096);
            place(sc.lds_media, (size_t)sc.n_media * sizeof(MediumRec), 2048);
            place(sc.lds_group_boxes, (size_t)sc.n_group_boxes * sizeof(GroupBox), 4096);
            place(sc.lds_materials, (size_t)sc.n_materials * sizeof(MaterialRec), 4096);
            if (T::RICH) place(sc.lds_perlin, (size_t)sc.n_perlin * sizeof(PerlinRec), 2 * sizeof(PerlinRec));
            const size_t b_quads = (size_t)sc.n_quads * sizeof(AAQuad), b_boxes = (size_t)sc.n_boxes * sizeof(BoxRec);
            if (T::BLOCK >= RT_BIG_BLOCK) {  // the big tables, most useful first
                place(sc.lds_boxes, b_boxes, 80 * 1024);
                place(sc.lds_spheres_tab, (size_t)sc.n_spheres * sizeof(SphereGeom), 40 * 1024);
                place(sc.lds_quad_aa, b_quads, 16 * 1024);
            } else if (b_quads + b_boxes <= 16 * 1024 && off + b_quads + b_boxes + 96 <= budget) {
                place(sc.lds_quad_aa, b_quads, 16 * 1024);
                place(sc.lds_boxes, b_boxes, 16 * 1024);
            }
            lds = off;
            if constexpr (T::BATCH && T::BLOCK >= RT_BIG_BLOCK) {
                // The deep kernel reads its node rows and every table from LDS only (its accessors have no global side: see
                // the head of render_kernel).  A scene that does not fit goes to the general kernel, which reads what is
                // not staged from L2.
                const bool fits = a.lds_nodes && (sc.n_objects == 0 || sc.lds_objects != kNone) && (sc.n_xforms == 0 || sc.lds_xforms != kNone) &&
                                  (sc.n_media == 0 || sc.lds_media != kNone) && (sc.n_group_boxes == 0 || sc.lds_group_boxes != kNone) &&
                                  (sc.n_materials == 0 || sc.lds_materials != kNone) && (sc.n_perlin == 0 || sc.lds_perlin != kNone) &&
                                  (sc.n_boxes == 0 || sc.lds_boxes != kNone) && (sc.n_spheres == 0 || sc.lds_spheres_tab != kNone);
                // (the quad rows stay optional: a box's six faces are read only for a hit point on one of its edges)
                if constexpr (T::SEG) {  // the reference's tree in the reference's order instead
                    if (!fits || !(sc.flags & SCENE_SEGMENTED) || sc.fast_nodes == nullptr || sc.n_seg_media > kSegMaxMedia)
                        return launch_one<TBvhGeneralDeep>(sc_in, a_in, stream, info);
                } else {
                    if (!fits) return launch_one<TBvhGeneral>(sc_in, a_in, stream, info);
                }
                auto empty = [](uint32_t &slot) { if (slot == kNone) slot = 0; };  // an empty table is never read
                empty(sc.lds_objects); empty(sc.lds_xforms); empty(sc.lds_media); empty(sc.lds_group_boxes); empty(sc.lds_materials);
                empty(sc.lds_perlin); empty(sc.lds_boxes); empty(sc.lds_spheres_tab);
            }
        }
    } else if (T::WORLD == 2) {
        lds = (T::BLOCK / 64) * kQueueCap * 64 * sizeof(uint16_t);
        size_t planes = (size_t)((sc.n_spheres + 63u) & ~63u) * 5 * sizeof(double);
        a.lds_spheres = 0;
        if (planes <= 48 * 1024) {
            lds += planes;
            a.lds_spheres = 1;
        }
    }
    if constexpr (T::PARK) {  // the parked path state, one entry per thread (list worlds stage no tables: their rows come through scalar loads)
        const size_t off = (lds + 15) & ~(size_t)15;
        sc.lds_park = (uint32_t)off;
        lds = off + (size_t)T::BLOCK * kParkBytesPerThread;
    }
    if (info) {
        hipFuncAttributes attr;
        hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kernel));
        if (e != hipSuccess) return e;
        info->vgprs = attr.numRegs;
        info->lds_bytes = (int)(attr.sharedSizeBytes + lds);
        info->kind = T::WORLD * 8 + (T::MEDIA ? 4 : 0) + (T::COMPOSITE ? 2 : 0) + (T::RICH ? 1 : 0) + (T::NESTED ? 32 : 0) + (T::FAST ? 64 : 0) +
                     (T::GROUPED ? 128 : 0) + (T::SEG ? 256 : 0);
        return hipSuccess;
    }
    if (a.n_pixels == 0 || a.spp <= 0) return hipSuccess;
    // persistent grid: as many workgroups as the chip holds at once (never more than there are tiles)
    int per_cu = 0;
    if (lds > 48 * 1024) {  // more dynamic LDS than the default limit: ask for it (up to the CU's 160 KB)
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return ea;
    }
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, T::BLOCK, lds);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    // Sphere-list worlds: two resident workgroups per CU beat three although three fit -- a third wave per SIMD
    // speeds the steady state up, but with fewer pixels per lane the frame tail grows by more (measured on C2).
    int cap = a.max_blocks_per_cu > 0 ? a.max_blocks_per_cu : (T::WORLD == 2 ? 2 : 0);
    if (cap > 0 && per_cu > cap) per_cu = cap;
    uint32_t resident = (uint32_t)per_cu * (uint32_t)(a.num_cus > 0 ? a.num_cus : 256);
    constexpr uint32_t kWavesPerBlock = (uint32_t)T::BLOCK / 64u;
    uint32_t blocks = (tiles + kWavesPerBlock - 1u) / kWavesPerBlock;
    if (a.pixel_list) blocks = resident;  // a pixel list's length lives on the device: the launcher caps the grid (grid_blocks)
    if (blocks > resident) blocks = resident;
    if (a.grid_blocks > 0 && blocks > (uint32_t)a.grid_blocks) blocks = (uint32_t)a.grid_blocks;  // tuning experiments
    dim3 grid(blocks), block(T::BLOCK);
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, sc, a);
    return hipGetLastError();
}

} // namespace

// instantiations of group 1, by id (defined in the RT_GROUP == 1 translation unit)
enum CompositeKernel { CK_LIST_PRIMS, CK_LIST_INSTANCES, CK_LIST_GENERAL, CK_LIST_NESTED, CK_BVH_INSTANCES, CK_BVH_MEDIA,
                       CK_BVH_GENERAL, CK_BVH_GENERAL_DEEP, CK_BVH_NESTED, CK_LIST_PRIMS_GROUPED, CK_LIST_INSTANCES_GROUPED, CK_BVH_SEGMENTED, CK_LIST_INSTANCES_5 };
hipError_t RT_CAT(launch_composite_, RT_SUFFIX)(int which, const DeviceScene &sc, const RenderArgs &a, hipStream_t stream, KernelInfo *info);

#if RT_GROUP == 1
hipError_t RT_CAT(launch_composite_, RT_SUFFIX)(int which, const DeviceScene &sc, const RenderArgs &a, hipStream_t stream, KernelInfo *info)
{
    switch (which) {
    case CK_LIST_PRIMS: return launch_one<TListPrims>(sc, a, stream, info);
    case CK_LIST_INSTANCES: return launch_one<TListInstances>(sc, a, stream, info);
    case CK_LIST_INSTANCES_5: return launch_one<TListInstances5>(sc, a, stream, info);
    case CK_LIST_PRIMS_GROUPED: return launch_one<TListPrimsGrouped>(sc, a, stream, info);
    case CK_LIST_INSTANCES_GROUPED: return launch_one<TListInstancesGrouped>(sc, a, stream, info);
    case CK_LIST_GENERAL: return launch_one<TListGeneral>(sc, a, stream, info);
    case CK_LIST_NESTED: return launch_one<TListNested>(sc, a, stream, info);
    case CK_BVH_INSTANCES: return launch_one<TBvhInstances>(sc, a, stream, info);
    case CK_BVH_MEDIA: return launch_one<TBvhMedia>(sc, a, stream, info);
    case CK_BVH_GENERAL: return launch_one<TBvhGeneral>(sc, a, stream, info);
    case CK_BVH_GENERAL_DEEP: return launch_one<TBvhGeneralDeep>(sc, a, stream, info);
    case CK_BVH_SEGMENTED: return launch_one<TBvhSegmented>(sc, a, stream, info);
    default: return launch_one<TBvhNested>(sc, a, stream, info);
    }
}
#else
namespace {
// Four or five waves per SIMD for the instanced-list kernel (TListInstances5): whole generations of pixels on the resident
// lanes times the duration of a pass at that occupancy (1 : 1.38, measured on C4: three generations of 85.7 ms against two of
// 118.6).  A frame that does not fill the lanes of four waves stays there: its time is its pixels' chains, and a pass is shortest
// with the fewest waves.
int list_instances_waves(const RenderArgs &a)
{
    if (a.list_waves == 4 || a.list_waves == 5) return a.list_waves;  // RT_TUNING builds / tests
    const double pixels = (double)a.width * (double)a.rows_owned;
    const double cus = a.num_cus > 0 ? (double)a.num_cus : 256.0;
    const double gen4 = std::ceil(pixels / (cus * 16.0 * 64.0) - 0.02), gen5 = std::ceil(pixels / (cus * 20.0 * 64.0) - 0.02);
    if (gen4 <= 1.0) return 4;
    return gen5 * 1.38 < gen4 ? 5 : 4;
}

hipError_t dispatch(const DeviceScene &sc, const RenderArgs &a, hipStream_t stream, KernelInfo *info)
{
    auto composite_kernel = [&](int which) { return RT_CAT(launch_composite_, RT_SUFFIX)(which, sc, a, stream, info); };
    const bool composite = sc.n_objects != 0 || sc.n_boxes != 0;
    const bool rich = (sc.flags & SCENE_RICH_TEXTURES) != 0;
    // RT_FLAG_ACCELERATE_LISTS: a list world of primitives through the library's tree, when its rows fit the kernel's LDS
    if (a.accelerate_lists && sc.world_kind == WORLD_LIST && sc.fast_nodes && !composite && !rich && !(sc.flags & SCENE_HAS_MEDIA) &&
        !(sc.flags & SCENE_HAS_TREES) && !a.force_general && fast_rows_fit(sc))
        return launch_one<TBvhPrimsFast>(sc, a, stream, info);
    if ((sc.flags & SCENE_LIST_ALL_SPHERES) && !rich && sc.n_spheres <= 65535u && !a.force_general)
        return launch_one<TSphereList>(sc, a, stream, info);
    if (sc.flags & SCENE_HAS_TREES) return composite_kernel(sc.world_kind == WORLD_BVH ? CK_BVH_NESTED : CK_LIST_NESTED);
    const bool media = (sc.flags & SCENE_HAS_MEDIA) != 0;
    const bool scan_world = sc.world_kind == WORLD_LIST || (sc.n_world_items <= 16u && sc.scan_cost <= (uint32_t)a.small_world && !a.always_walk);
    if (scan_world && !rich && !media && !a.force_general) {
        // pixels_per_wave < 64 (a power of two: rt_render_launch): the instantiation that deals a ray's leaves to lanes
        const bool grouped = a.pixels_per_wave < 64 && (a.pixels_per_wave & (a.pixels_per_wave - 1)) == 0 && a.pixels_per_wave > 0;
        if (grouped) return composite_kernel(composite ? CK_LIST_INSTANCES_GROUPED : CK_LIST_PRIMS_GROUPED);
        if (composite && list_instances_waves(a) == 5) return composite_kernel(CK_LIST_INSTANCES_5);
        return composite_kernel(composite ? CK_LIST_INSTANCES : CK_LIST_PRIMS);
    }
    if (sc.world_kind == WORLD_BVH) {
        if (!composite && !rich && !a.force_general)
            return (sc.fast_nodes && !a.reference_tree && sc.n_fast_nodes * kFastNodeBytes <= kFastLdsBudget)
                       ? launch_one<TBvhPrimsFast>(sc, a, stream, info)
                       : launch_one<TBvhPrims>(sc, a, stream, info);
        if (!rich && !a.force_general) return composite_kernel(media ? CK_BVH_MEDIA : CK_BVH_INSTANCES);
        // deep worlds: the library's tree, one walk per run of surfaces between media, where the scene has one (RT_FLAG_REFERENCE_TREE:
        // the reference's tree in the reference's order); both fall back when their tables do not fit the LDS of a CU
        if (sc.n_world_nodes > 64 && (sc.flags & SCENE_SEGMENTED) && !a.reference_tree) return composite_kernel(CK_BVH_SEGMENTED);
        return composite_kernel(sc.n_world_nodes > 64 ? CK_BVH_GENERAL_DEEP : CK_BVH_GENERAL);
    }
    return composite_kernel(CK_LIST_GENERAL);
}
} // namespace

hipError_t RT_CAT(launch_render_, RT_SUFFIX)(const DeviceScene &sc, const RenderArgs &a, hipStream_t stream)
{
    return dispatch(sc, a, stream, nullptr);
}

hipError_t RT_CAT(kernel_info_, RT_SUFFIX)(const DeviceScene &sc, const RenderArgs &a, KernelInfo *info)
{
    return dispatch(sc, a, nullptr, info);
}
#endif

} // namespace rtow
