// render.hip -- the gfx950 path-tracing megakernel and the RNG seeding kernel.
//
// Replaces the reference's RenderInit + Render kernels (R/kernel.cu:110-154) and everything they
// inline: Camera::GetRay (R/Camera.h:76-85), RayColor (R/kernel.cu:65-98), BvhNode::Hit
// (R/BvhNode.h:101-158), the primitive / instance / medium Hit functions, Material::Scatter/Emitted
// and Texture::Value.  Not a translation: one lane owns one pixel and runs a flat
// "one ray segment per iteration" loop with path regeneration (a lane whose path ends starts its
// pixel's next sample in the same iteration), virtual dispatch is tag dispatch over the SoA tables
// of flat_scene.h, the BVH is walked stacklessly through escape links, hit records are built once
// per bounce from (t, primitive) instead of on every accepted candidate, and sphere UVs are computed
// only when an image texture will read them.  Each of these is result-preserving: see DESIGN.md.
//
// This file is compiled twice: RT_STRICT=1 with -ffp-contract=off (no FMA; bit-comparable with the
// CPU oracle) and RT_STRICT=0 with the default contraction (fast variant).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "flat_scene.h"
#include "render_iface.h"
#include "rng.h"

#ifndef RT_STRICT
#define RT_STRICT 1
#endif

namespace rtow {
namespace {

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

#define DEV __device__ __forceinline__

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
DEV Vec load3(const double *p) { return mk(p[0], p[1], p[2]); }

// ------------------------------------------------------------------------------------------------
// primitive tests.  Each returns the accepted t exactly as the reference's Hit would set rec.T.
// ------------------------------------------------------------------------------------------------
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
        double s = sqrt(disc);
        double t = (-b - s) / a;
        if (t < tmax && t > tmin) {
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

DEV Vec msphere_center(const MSphereGeom &g, double tm)  // R/MovingSphere.h:51-52
{
    double frac = (tm - g.t0) / g.dt;
    return mk(g.c0x, g.c0y, g.c0z) + frac * mk(g.dcx, g.dcy, g.dcz);
}

// Test one primitive reference; on success updates closest/best.
DEV bool prim_test(const DeviceScene &sc, uint32_t ref, const Ray &r, double a, double tmin, double tmax, double &t)
{
    uint32_t idx = ref & kRefIndexMask;
    switch (ref >> kRefShift) {
    case REF_SPHERE: {
        SphereGeom g = sc.spheres[idx];
        return sphere_test(r.o - mk(g.cx, g.cy, g.cz), r.d, a, g.r2, tmin, tmax, t);
    }
    case REF_MSPHERE: {
        MSphereGeom g = sc.mspheres[idx];
        return sphere_test(r.o - msphere_center(g, r.tm), r.d, a, g.r2, tmin, tmax, t);
    }
    default: {
        return quad_test(sc.quads[idx], r, tmin, tmax, t);
    }
    }
}

// Ray into the object space of a composite leaf: Translate (R/Instance.h:46) and RotateY (:121-131)
// applied outermost first.
DEV Ray to_object_space(const DeviceScene &sc, const ObjectRec &o, const Ray &r)
{
    Ray lr = r;
    for (uint32_t k = 0; k < o.xf_count; k++) {
        Xform x = sc.xforms[o.xf_first + k];
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

// Closest hit over a composite leaf's geometry (R/HittableList.h:39-57 for groups).
DEV bool geom_closest(const DeviceScene &sc, const ObjectRec &o, const Ray &lr, double tmin, double tmax,
                      double &t_best, uint32_t &ref_best)
{
    double a = dot(lr.d, lr.d);
    bool any = false;
    double closest = tmax;
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
            SphereGeom g = sc.spheres[o.first + k];
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
    case GEOM_QUADS:
        for (uint32_t k = 0; k < o.count; k++) {
            double t;
            if (quad_test(sc.quads[o.first + k], lr, tmin, closest, t)) {
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
DEV bool object_test(const DeviceScene &sc, uint32_t oi, const Ray &r, double tmin, double tmax, HitInfo &best, Xorwow &rng)
{
    ObjectRec o = sc.objects[oi];
    Ray lr = to_object_space(sc, o, r);
    if (o.medium == kNone) {
        double t;
        uint32_t pref;
        if (!geom_closest(sc, o, lr, tmin, tmax, t, pref)) return false;
        best.t = t;
        best.ref = pref;
        best.obj = oi;
        return true;
    }
    double t1, t2;
    uint32_t unused;
    if (!geom_closest(sc, o, lr, -DBL_MAX, DBL_MAX, t1, unused)) return false;
    if (!geom_closest(sc, o, lr, t1 + 0.0001, DBL_MAX, t2, unused)) return false;
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return false;
    if (t1 < 0.0) t1 = 0.0;
    double ray_len = length(r.d);
    double inside = (t2 - t1) * ray_len;
    // log(curand_uniform(..)) has a float argument: the float overload is selected on the reference's
    // toolchain; evaluated here as the correctly rounded fp32 log.
    float lg = (float)log((double)xorwow_uniform(rng));
    double hit_dist = sc.media[o.medium].neg_inv_density * (double)lg;
    if (hit_dist > inside) return false;
    best.t = t1 + hit_dist / ray_len;
    best.ref = make_ref(REF_MEDIUM, o.medium);
    best.obj = oi;
    return true;
}

DEV bool leaf_test(const DeviceScene &sc, uint32_t ref, const Ray &r, double a, double tmin, double tmax, HitInfo &best, Xorwow &rng)
{
    if ((ref >> kRefShift) == REF_OBJECT) return object_test(sc, ref & kRefIndexMask, r, tmin, tmax, best, rng);
    double t;
    if (!prim_test(sc, ref, r, a, tmin, tmax, t)) return false;
    best.t = t;
    best.ref = ref;
    best.obj = kNone;
    return true;
}

DEV bool is_medium_leaf(const DeviceScene &sc, uint32_t ref)
{
    return (ref >> kRefShift) == REF_OBJECT && sc.objects[ref & kRefIndexMask].medium != kNone;
}

// ------------------------------------------------------------------------------------------------
// world traversal
// ------------------------------------------------------------------------------------------------
// Slab test, R/AABB.h:68-98, with 1/d hoisted out of the node loop (same value every time).
DEV bool box_test(const BvhNodeRec &n, const Ray &r, Vec inv, double tmin, double tmax)
{
    double t0 = (n.xlo - r.o.x) * inv.x, t1 = (n.xhi - r.o.x) * inv.x;
    tmin = fmax(tmin, fmin(t0, t1));
    tmax = fmin(tmax, fmax(t0, t1));
    t0 = (n.ylo - r.o.y) * inv.y;
    t1 = (n.yhi - r.o.y) * inv.y;
    tmin = fmax(tmin, fmin(t0, t1));
    tmax = fmin(tmax, fmax(t0, t1));
    t0 = (n.zlo - r.o.z) * inv.z;
    t1 = (n.zhi - r.o.z) * inv.z;
    tmin = fmax(tmin, fmin(t0, t1));
    tmax = fmin(tmax, fmax(t0, t1));
    return tmax > tmin;
}

// Stackless walk in the reference's visiting order (R/BvhNode.h:101-158): a node's leaf children are
// tested where the node is visited; "pop" is the escape link.
DEV bool world_hit_bvh(const DeviceScene &sc, const Ray &r, double tmin, double tmax, HitInfo &best, Xorwow &rng)
{
    Vec inv = mk(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    double a = dot(r.d, r.d);
    double closest = tmax;
    bool any = false;
    uint32_t n = 0;
    while (n != kNone) {
        BvhNodeRec node = sc.nodes[n];
        uint32_t next = node.escape;
        if (box_test(node, r, inv, tmin, closest)) {
            if ((node.a >> kRefShift) == REF_INNER) {
                next = n + 1;
            } else {
                if (leaf_test(sc, node.a, r, a, tmin, closest, best, rng)) {
                    any = true;
                    closest = best.t;
                }
                // span-1 nodes hold the same leaf twice (R/BvhNode.h:63-67).  Re-testing a surface with
                // tmax = its own t changes nothing; a medium draws again, so only media are re-tested.
                if (node.b != node.a || is_medium_leaf(sc, node.b)) {
                    if (leaf_test(sc, node.b, r, a, tmin, closest, best, rng)) {
                        any = true;
                        closest = best.t;
                    }
                }
            }
        }
        n = next;
    }
    return any;
}

// HittableList world (R/HittableList.h:39-57): the item index is wave-uniform, so the primitive rows
// are fetched through the scalar path.
DEV bool world_hit_list(const DeviceScene &sc, const Ray &r, double tmin, double tmax, HitInfo &best, Xorwow &rng)
{
    double a = dot(r.d, r.d);
    double closest = tmax;
    bool any = false;
    if (sc.flags & SCENE_LIST_ALL_SPHERES) {
        const uint32_t n = sc.n_spheres;
        for (uint32_t k = 0; k < n; k++) {
            SphereGeom g = sc.spheres[k];
            double t;
            if (sphere_test(r.o - mk(g.cx, g.cy, g.cz), r.d, a, g.r2, tmin, closest, t)) {
                any = true;
                closest = t;
                best.t = t;
                best.ref = make_ref(REF_SPHERE, k);
                best.obj = kNone;
            }
        }
        return any;
    }
    const uint32_t n = sc.n_world_items;
    for (uint32_t k = 0; k < n; k++) {
        uint32_t ref = sc.world_items[k];
        if (leaf_test(sc, ref, r, a, tmin, closest, best, rng)) {
            any = true;
            closest = best.t;
        }
    }
    return any;
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

DEV Surface make_surface(const DeviceScene &sc, const Ray &r, const HitInfo &h)
{
    Surface s;
    s.u = 0.0;
    s.v = 0.0;
    uint32_t tag = h.ref >> kRefShift, idx = h.ref & kRefIndexMask;
    if (tag == REF_MEDIUM) {  // R/ConstantMedium.h:86-91
        s.p = at(r, h.t);
        s.n = mk(1, 0, 0);
        s.front = true;
        s.mat = sc.media[idx].phase_mat;
        return s;
    }
    ObjectRec o;
    Ray lr = r;
    if (h.obj != kNone) {
        o = sc.objects[h.obj];
        lr = to_object_space(sc, o, r);
    }
    s.p = at(lr, h.t);
    if (tag == REF_QUAD) {  // R/Quad.h:86-96
        QuadGeom q = sc.quads[idx];
        s.mat = sc.quad_mat[idx];
        face(s, lr, mk(q.nx, q.ny, q.nz));
        if (sc.materials[s.mat].needs_uv) {
            Vec ph = s.p - mk(q.qx, q.qy, q.qz);
            Vec w = mk(q.wx, q.wy, q.wz);
            s.u = dot(w, cross(ph, mk(q.vx, q.vy, q.vz)));
            s.v = dot(w, cross(mk(q.ux, q.uy, q.uz), ph));
        }
    } else {  // R/Sphere.h:40-46
        Vec c;
        SphereAux aux;
        if (tag == REF_SPHERE) {
            SphereGeom g = sc.spheres[idx];
            c = mk(g.cx, g.cy, g.cz);
            aux = sc.sphere_aux[idx];
        } else {
            c = msphere_center(sc.mspheres[idx], lr.tm);
            aux = sc.msphere_aux[idx];
        }
        Vec on = aux.inv_r * (s.p - c);
        face(s, lr, on);
        s.mat = aux.mat;
        if (sc.materials[s.mat].needs_uv) sphere_uv(on, s.u, s.v);
    }
    if (h.obj != kNone) {  // back to world space, innermost transform first (R/Instance.h:53,137-147)
        for (uint32_t k = o.xf_count; k-- > 0;) {
            Xform x = sc.xforms[o.xf_first + k];
            if (x.kind == XF_TRANSLATE) {
                s.p = s.p + mk(x.a, x.b, x.c);
            } else {
                double st = x.a, ct = x.b;
                s.p = mk((ct * s.p.x) + (st * s.p.z), s.p.y, (-st * s.p.x) + (ct * s.p.z));
                s.n = mk((ct * s.n.x) + (st * s.n.z), s.n.y, (-st * s.n.x) + (ct * s.n.z));
            }
        }
    }
    return s;
}

// ------------------------------------------------------------------------------------------------
// textures and materials
// ------------------------------------------------------------------------------------------------
DEV double perlin_noise(const PerlinRec &pn, Vec p)  // R/Perlin.h:38-60,120-139
{
    double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    double uu = u * u * (3.0 - 2.0 * u), vv = v * v * (3.0 - 2.0 * v), ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                int idx = pn.perm_x[(i + a) & 255] ^ pn.perm_y[(j + b) & 255] ^ pn.perm_z[(k + c) & 255];
                Vec g = mk(pn.vec[idx][0], pn.vec[idx][1], pn.vec[idx][2]);
                Vec wv = mk(u - a, v - b, w - c);
                accum += (a * uu + (1 - a) * (1 - uu)) * (b * vv + (1 - b) * (1 - vv)) * (c * ww + (1 - c) * (1 - ww)) * dot(g, wv);
            }
    return accum;
}

DEV double perlin_turb(const PerlinRec &pn, Vec p, int depth)  // R/Perlin.h:63-78
{
    double accum = 0.0, weight = 1.0;
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(pn, p);
        weight *= 0.5;
        p = 2.0 * p;
    }
    return fabs(accum);
}

DEV Vec texture_value(const DeviceScene &sc, uint32_t ti, double u, double v, Vec p)
{
    TextureRec t = sc.textures[ti];
    while (t.kind == TEX_CHECKER) {  // R/Texture.h:70-81
        int xi = (int)floor(t.s * p.x), yi = (int)floor(t.s * p.y), zi = (int)floor(t.s * p.z);
        bool even = ((xi + yi + zi) % 2) == 0;
        t = sc.textures[even ? t.a : t.b_];
    }
    if (t.kind == TEX_SOLID) return mk(t.r, t.g, t.b);
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
    // R/Texture.h:159-165: marble
    double sv = 1.0 + sin(t.s * p.z + 10.0 * perlin_turb(sc.perlin[t.a], p, 7));
    return sv * mk(0.5, 0.5, 0.5);
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
DEV bool shade(const DeviceScene &sc, const Surface &s, Ray &ray, Vec &throughput, Vec &accumulated, Xorwow &rng)
{
    MaterialRec m = sc.materials[s.mat];
    Vec atten;
    Ray out;
    out.o = s.p;
    out.tm = ray.tm;
    switch (m.kind) {
    case MAT_DIFFUSE_LIGHT:  // R/Material.h:114-127: emits on both sides, never scatters
        accumulated = accumulated + throughput * texture_value(sc, m.tex, s.u, s.v, s.p);
        return false;
    case MAT_LAMBERTIAN: {  // R/Material.h:67-82
        Vec dir = s.n + random_in_unit_sphere(rng);
        if (fabs(dir.x) < 1e-8 && fabs(dir.y) < 1e-8 && fabs(dir.z) < 1e-8) dir = s.n;
        out.d = dir;
        atten = texture_value(sc, m.tex, s.u, s.v, s.p);
        break;
    }
    case MAT_METAL: {  // R/Metal.h:18-30
        Vec refl = reflect(unit(ray.d), s.n);
        out.d = refl + m.p * random_in_unit_sphere(rng);
        atten = mk(m.r, m.g, m.b);
        if (!(dot(out.d, s.n) > 0.0)) return false;
        break;
    }
    case MAT_DIELECTRIC: {  // R/Dielectric.h:18-68
        atten = mk(1.0, 1.0, 1.0);
        double ratio = s.front ? (1.0 / m.p) : m.p;
        Vec ud = unit(ray.d);
        double ct = fmin(dot(-ud, s.n), 1.0);
        double st = sqrt(1.0 - ct * ct);
        bool reflect_it = ratio * st > 1.0;
        if (!reflect_it) {
            double r0 = (1.0 - ratio) / (1.0 + ratio);
            r0 = r0 * r0;
            double refl = r0 + (1.0 - r0) * pow(1.0 - ct, 5.0);
            reflect_it = refl > (double)xorwow_uniform(rng);
        }
        out.d = reflect_it ? reflect(ud, s.n) : refract(ud, s.n, ratio);
        break;
    }
    default: {  // MAT_ISOTROPIC, R/Material.h:152-163
        out.d = unit(random_in_unit_sphere(rng));
        atten = texture_value(sc, m.tex, s.u, s.v, s.p);
        break;
    }
    }
    throughput = throughput * atten;
    ray = out;
    return true;
}

// Camera::GetRay (R/Camera.h:76-85) behind the pixel jitter of Render (R/kernel.cu:140-142).
DEV Ray camera_ray(const CameraRec &cam, int i, int j, int width, int height, Xorwow &rng)
{
    double u = (double)((float)i + xorwow_uniform(rng)) / (double)width;   // int + float adds in fp32
    double v = (double)((float)j + xorwow_uniform(rng)) / (double)height;
    Vec p;
    do {
        double a = (double)xorwow_uniform(rng);
        double b = (double)xorwow_uniform(rng);
        p = 2.0 * mk(a, b, 0.0) - mk(1.0, 1.0, 0.0);
    } while (dot(p, p) >= 1.0);
    Vec rd = cam.lens_radius * p;
    Vec cu = load3(cam.u), cv = load3(cam.v);
    Vec offset = rd.x * cu + rd.y * cv;
    double tm = cam.time0 + (double)xorwow_uniform(rng) * (cam.time1 - cam.time0);
    Vec origin = load3(cam.origin);
    Ray r;
    r.o = origin + offset;
    r.d = load3(cam.llc) + u * load3(cam.horizontal) + v * load3(cam.vertical) - origin - offset;
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

template <int STRICT>
__global__ __launch_bounds__(256) void render_kernel(DeviceScene sc, RenderArgs a)
{
    // one wave = one 8x8 pixel tile of this rank's compact image
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t tiles_x = ((uint32_t)a.width + 7u) >> 3;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const int i = (int)(tx * 8u + (lane & 7u));
    const int lr = (int)(ty * 8u + (lane >> 3));
    bool active = i < a.width && lr < a.rows_owned && a.spp > 0;
    const int j = owned_row(lr, a.stripe_rows, a.rank, a.world_size);
    const size_t local = (size_t)lr * (size_t)a.width + (size_t)i;

    Xorwow rng{0, 0, 0, 0, 0, 0};
    if (active) {
        rng.d = a.state[0 * (size_t)a.n_pixels + local];
        rng.v0 = a.state[1 * (size_t)a.n_pixels + local];
        rng.v1 = a.state[2 * (size_t)a.n_pixels + local];
        rng.v2 = a.state[3 * (size_t)a.n_pixels + local];
        rng.v3 = a.state[4 * (size_t)a.n_pixels + local];
        rng.v4 = a.state[5 * (size_t)a.n_pixels + local];
    }

    const Vec background = load3(sc.camera.bg);
    Vec col = mk(0.0, 0.0, 0.0);
    Vec throughput = mk(1.0, 1.0, 1.0), accumulated = mk(0.0, 0.0, 0.0);
    Ray ray{};
    int sample = 0, depth = 0;
    unsigned long long nrays = 0;
    if (active) ray = camera_ray(sc.camera, i, j, a.width, a.height, rng);

    // Flat loop: every iteration traces one ray segment for every lane that still has samples left.
    while (active) {
        HitInfo h;
        h.t = 0.0;
        h.ref = kNone;
        h.obj = kNone;
        nrays++;
        bool hit = (sc.world_kind == WORLD_BVH) ? world_hit_bvh(sc, ray, 0.001, DBL_MAX, h, rng)
                                                 : world_hit_list(sc, ray, 0.001, DBL_MAX, h, rng);
        bool path_ends;
        if (!hit) {  // R/kernel.cu:74-79
            accumulated = accumulated + throughput * background;
            path_ends = true;
        } else {
            Surface s = make_surface(sc, ray, h);
            path_ends = !shade(sc, s, ray, throughput, accumulated, rng);
            if (!path_ends && ++depth >= a.max_depth) path_ends = true;  // R/kernel.cu:71,97
        }
        if (path_ends) {  // R/kernel.cu:143: col += RayColor(...)
            col = col + accumulated;
            if (++sample < a.spp) {
                ray = camera_ray(sc.camera, i, j, a.width, a.height, rng);
                throughput = mk(1.0, 1.0, 1.0);
                accumulated = mk(0.0, 0.0, 0.0);
                depth = 0;
            } else {
                active = false;
            }
        }
    }

    if (i < a.width && lr < a.rows_owned) {
        // R/kernel.cu:146-153: save the RNG state, average, gamma 2
        a.state[0 * (size_t)a.n_pixels + local] = rng.d;
        a.state[1 * (size_t)a.n_pixels + local] = rng.v0;
        a.state[2 * (size_t)a.n_pixels + local] = rng.v1;
        a.state[3 * (size_t)a.n_pixels + local] = rng.v2;
        a.state[4 * (size_t)a.n_pixels + local] = rng.v3;
        a.state[5 * (size_t)a.n_pixels + local] = rng.v4;
        if (a.spp > 0) {
            col = over(col, (double)a.spp);
            a.pixels[local * 3 + 0] = sqrt(col.x);
            a.pixels[local * 3 + 1] = sqrt(col.y);
            a.pixels[local * 3 + 2] = sqrt(col.z);
        }
    }
    // one atomic per wave for the ray counter
    for (int off = 32; off > 0; off >>= 1) nrays += __shfl_down(nrays, off, 64);
    if (lane == 0 && nrays) atomicAdd(a.ray_counter, nrays);
}

#if RT_STRICT
#define RT_SUFFIX strict
#else
#define RT_SUFFIX fast
#endif
#define RT_CAT2(a, b) a##b
#define RT_CAT(a, b) RT_CAT2(a, b)

hipError_t RT_CAT(launch_seed_, RT_SUFFIX)(const SeedArgs &a, hipStream_t stream)
{
    if (a.n_pixels == 0) return hipSuccess;
    dim3 grid((a.n_pixels + 255u) / 256u), block(256);
    hipLaunchKernelGGL(seed_kernel<RT_STRICT>, grid, block, 0, stream, a);
    return hipGetLastError();
}

hipError_t RT_CAT(launch_render_, RT_SUFFIX)(const DeviceScene &sc, const RenderArgs &a, hipStream_t stream)
{
    if (a.n_pixels == 0) return hipSuccess;
    uint32_t tiles = (((uint32_t)a.width + 7u) >> 3) * (((uint32_t)a.rows_owned + 7u) >> 3);
    dim3 grid((tiles + 3u) / 4u), block(256);
    hipLaunchKernelGGL(render_kernel<RT_STRICT>, grid, block, 0, stream, sc, a);
    return hipGetLastError();
}

hipError_t RT_CAT(kernel_attributes_, RT_SUFFIX)(int *vgprs, int *lds_bytes)
{
    hipFuncAttributes attr;
    hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&render_kernel<RT_STRICT>));
    if (e != hipSuccess) return e;
    *vgprs = attr.numRegs;
    *lds_bytes = (int)attr.sharedSizeBytes;
    return hipSuccess;
}

} // namespace rtow
