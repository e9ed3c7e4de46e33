// scene_builder.cpp -- host side of the construction API (include/rtow.h) and the flattener.
//
// The reference builds its world on the device with `new` inside a <<<1,1>>> kernel
// (R/kernel.cu:176-543).  Here the same constructors run on the host, keep a handle-addressed object
// graph, and rt_scene_commit() lowers that graph to the SoA tables of flat_scene.h.  All arithmetic
// that decides geometry (bounding boxes, quad planes, rotation constants, BVH split order) follows
// the reference expression by expression, in fp64 without contraction, so the tables are bit-identical
// to what the reference's constructors compute.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <limits>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "../../include/rtow.h"
#include "scene_host.h"

namespace rtow {

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
int fail(int status, const std::string &msg)
{
    g_error = msg;
    return status;
}

// ------------------------------------------------------------------------------------------------
// small vector helpers (R/Vec3.h semantics: v / t is (1 / t) * v)
// ------------------------------------------------------------------------------------------------
static inline D3 mk(double x, double y, double z) { return D3{x, y, z}; }
static inline D3 add(D3 a, D3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline D3 sub(D3 a, D3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline D3 neg(D3 a) { return mk(-a.x, -a.y, -a.z); }
static inline D3 scale(double t, D3 a) { return mk(t * a.x, t * a.y, t * a.z); }
static inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline D3 cross(D3 u, D3 v) { return mk(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
static inline double length(D3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline D3 over(D3 a, double t) { return scale(1 / t, a); }
static inline D3 normalize(D3 a) { return over(a, length(a)); }
static inline double comp(D3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

// ------------------------------------------------------------------------------------------------
// boxes (R/AABB.h, R/Interval.h)
// ------------------------------------------------------------------------------------------------
static Box empty_box()
{
    Box b;
    for (int k = 0; k < 3; k++) {
        b.lo[k] = +DBL_MAX;
        b.hi[k] = -DBL_MAX;
    }
    return b;
}
static void pad_thin_axes(Box &b)  // AABB.h:114-120
{
    const double delta = 0.0001;
    for (int k = 0; k < 3; k++)
        if (b.hi[k] - b.lo[k] < delta) {
            double half = delta / 2.0;
            b.lo[k] = b.lo[k] - half;
            b.hi[k] = b.hi[k] + half;
        }
}
static Box box_from_corners(D3 a, D3 b)  // AABB.h:34-40
{
    Box r;
    for (int k = 0; k < 3; k++) {
        double p = comp(a, k), q = comp(b, k);
        if (p <= q) {
            r.lo[k] = p;
            r.hi[k] = q;
        } else {
            r.lo[k] = q;
            r.hi[k] = p;
        }
    }
    pad_thin_axes(r);
    return r;
}
static Box box_merge(const Box &a, const Box &b)  // AABB.h:43-48 (no padding)
{
    Box r;
    for (int k = 0; k < 3; k++) {
        r.lo[k] = a.lo[k] <= b.lo[k] ? a.lo[k] : b.lo[k];
        r.hi[k] = a.hi[k] >= b.hi[k] ? a.hi[k] : b.hi[k];
    }
    return r;
}
static Box box_moved(const Box &b, D3 off)  // AABB.h:127-130 (interval ctor pads again)
{
    Box r;
    for (int k = 0; k < 3; k++) {
        r.lo[k] = b.lo[k] + comp(off, k);
        r.hi[k] = b.hi[k] + comp(off, k);
    }
    pad_thin_axes(r);
    return r;
}
static int longest_axis(const Box &b)  // AABB.h:101-107
{
    double sx = b.hi[0] - b.lo[0], sy = b.hi[1] - b.lo[1], sz = b.hi[2] - b.lo[2];
    if (sx > sy) return sx > sz ? 0 : 2;
    return sy > sz ? 1 : 2;
}

// ------------------------------------------------------------------------------------------------
// host RNG jump table: T^(2^67 * g * 16^k) for g = 1..15, k = 0..15
// ------------------------------------------------------------------------------------------------
namespace {
struct Gf2 {
    uint32_t row[160][5];
};
void gf2_mul_vec(const Gf2 &m, const uint32_t in[5], uint32_t out[5])
{
    uint32_t acc[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 160; i++)
        if ((in[i >> 5] >> (i & 31)) & 1u)
            for (int k = 0; k < 5; k++) acc[k] ^= m.row[i][k];
    std::memcpy(out, acc, sizeof acc);
}
// r = a after b (apply b first, then a): row_i(r) = a(row_i(b))
void gf2_compose(const Gf2 &a, const Gf2 &b, Gf2 &r)
{
    for (int i = 0; i < 160; i++) gf2_mul_vec(a, b.row[i], r.row[i]);
}
std::vector<uint32_t> build_jump_table()
{
    std::vector<uint32_t> table(kJumpTableWords);
    Gf2 *cur = new Gf2, *tmp = new Gf2, *pw = new Gf2;
    for (int i = 0; i < 160; i++) {
        Xorwow s{0, 0, 0, 0, 0, 0};
        uint32_t *w = &s.v0;
        w[i >> 5] = 1u << (i & 31);
        xorwow_next(s);
        cur->row[i][0] = s.v0; cur->row[i][1] = s.v1; cur->row[i][2] = s.v2; cur->row[i][3] = s.v3; cur->row[i][4] = s.v4;
    }
    for (int k = 0; k < 67; k++) {  // T^(2^67)
        gf2_compose(*cur, *cur, *tmp);
        std::swap(cur, tmp);
    }
    for (int k = 0; k < kJumpDigits; k++) {
        // cur = J^(16^k); fill g = 1..15 by repeated composition
        *pw = *cur;
        for (int g = 1; g <= 15; g++) {
            std::memcpy(&table[((size_t)k * 15 + (g - 1)) * kJumpMatrixWords], pw->row, sizeof pw->row);
            if (g < 15) {
                gf2_compose(*cur, *pw, *tmp);
                *pw = *tmp;
            }
        }
        gf2_compose(*cur, *pw, *tmp);  // J^(16^k * 16)
        *cur = *tmp;
    }
    delete cur;
    delete tmp;
    delete pw;
    return table;
}
} // namespace

const uint32_t *host_jump_table()
{
    static std::once_flag once;
    static std::vector<uint32_t> table;
    std::call_once(once, [] { table = build_jump_table(); });
    return table.data();
}

// ------------------------------------------------------------------------------------------------
// graph access helpers
// ------------------------------------------------------------------------------------------------
static HostHittable *get_h(SceneImpl *s, rt_handle h)
{
    if (!s || h == 0 || h > s->hittables.size()) return nullptr;
    return &s->hittables[h - 1];
}
static bool valid_mat(SceneImpl *s, rt_handle m) { return s && m >= 1 && m <= s->materials.size(); }
static bool valid_tex(SceneImpl *s, rt_handle t) { return s && t >= 1 && t <= s->textures.size(); }
static rt_handle push_h(SceneImpl *s, HostHittable &&h)
{
    s->committed = false;
    s->hittables.push_back(std::move(h));
    return (rt_handle)s->hittables.size();
}
static rt_handle push_tex(SceneImpl *s, const HostTexture &t)
{
    s->committed = false;
    s->textures.push_back(t);
    return (rt_handle)s->textures.size();
}
static rt_handle push_mat(SceneImpl *s, const HostMaterial &m)
{
    s->committed = false;
    s->materials.push_back(m);
    return (rt_handle)s->materials.size();
}

// ------------------------------------------------------------------------------------------------
// BVH construction: R/BvhNode.h:50-90 (recursive median split) + :170-193 (stable insertion sort)
// ------------------------------------------------------------------------------------------------
static int build_tree(SceneImpl *s, HostHittable &bvh, std::vector<uint32_t> &objs, int start, int end)
{
    int me = (int)bvh.tree.size();
    bvh.tree.push_back({});
    Box box = empty_box();
    for (int i = start; i < end; i++) box = box_merge(box, s->hittables[objs[i] - 1].box);
    int axis = longest_axis(box);
    int span = end - start;
    HostHittable::TreeNode node{};
    node.box = box;
    node.left = node.right = -1;
    if (span == 1) {
        node.leaf_a = node.leaf_b = objs[start];
    } else if (span == 2) {
        node.leaf_a = objs[start];
        node.leaf_b = objs[start + 1];
    } else {
        for (int i = start + 1; i < end; i++) {
            uint32_t key = objs[i];
            double key_min = s->hittables[key - 1].box.lo[axis];
            int j = i - 1;
            while (j >= start && key_min < s->hittables[objs[j] - 1].box.lo[axis]) {
                objs[j + 1] = objs[j];
                j--;
            }
            objs[j + 1] = key;
        }
        int mid = start + span / 2;
        node.left = build_tree(s, bvh, objs, start, mid);
        node.right = build_tree(s, bvh, objs, mid, end);
    }
    bvh.tree[me] = node;
    return me;
}

// ------------------------------------------------------------------------------------------------
// flattening
// ------------------------------------------------------------------------------------------------
namespace {
// Coincident primitives: two identical spheres, or two quads in one plane whose rectangles overlap.  A ray that hits both
// gets the same t twice, and then the ORDER of the tests decides which material it sees: the reference's list keeps the
// first sphere it meets (strict `<`, R/Sphere.h:38,50) and the last quad (inclusive interval, R/Quad.h:59-64).  The
// library's own accelerators (the near-child-first tree of a primitive world, the sub-BVH / cooperative scan of a large
// group) meet the primitives in another order, so they are not built over such a set: it keeps the reference's tree or
// its linear list.  Exact comparisons on purpose: surfaces that differ in the last bit do not tie.
static bool has_coincident_primitives(const SceneImpl &s, const std::vector<uint32_t> &handles)
{
    struct Key {
        double v[10];
        uint32_t handle;
    };
    auto less = [](const Key &a, const Key &b) { return std::lexicographical_compare(a.v, a.v + 10, b.v, b.v + 10); };
    auto same = [](const Key &a, const Key &b) { return std::equal(a.v, a.v + 10, b.v); };
    std::vector<Key> spheres, planes;
    for (uint32_t hnd : handles) {
        const HostHittable &h = s.hittables[hnd - 1];
        if (h.kind == HKind::Sphere || h.kind == HKind::MovingSphere) {
            const bool moving = h.kind == HKind::MovingSphere;
            spheres.push_back({{h.c0.x, h.c0.y, h.c0.z, moving ? h.c1.x : h.c0.x, moving ? h.c1.y : h.c0.y, moving ? h.c1.z : h.c0.z,
                                moving ? h.t0 : 0.0, moving ? h.t1 : 0.0, h.radius, moving ? 1.0 : 0.0}, hnd});
        } else if (h.kind == HKind::Quad) {
            // the plane, with the sign of the normal fixed by its first non-zero component
            double n[3] = {h.normal.x, h.normal.y, h.normal.z}, d = h.plane_d;
            const double lead = n[0] != 0.0 ? n[0] : (n[1] != 0.0 ? n[1] : n[2]);
            if (lead < 0.0) {
                for (double &c : n) c = -c;
                d = -d;
            }
            planes.push_back({{n[0] + 0.0, n[1] + 0.0, n[2] + 0.0, d + 0.0, 0, 0, 0, 0, 0, 0}, hnd});  // + 0.0: -0.0 -> +0.0
        }
    }
    std::sort(spheres.begin(), spheres.end(), less);
    for (size_t k = 1; k < spheres.size(); k++)
        if (same(spheres[k - 1], spheres[k])) return true;
    std::sort(planes.begin(), planes.end(), less);
    for (size_t a = 0; a < planes.size(); a++)
        for (size_t b = a + 1; b < planes.size() && same(planes[a], planes[b]); b++) {
            const Box &x = s.hittables[planes[a].handle - 1].box, &y = s.hittables[planes[b].handle - 1].box;
            bool overlap = true;
            for (int k = 0; k < 3; k++) overlap &= x.lo[k] <= y.hi[k] && y.lo[k] <= x.hi[k];
            if (overlap) return true;
        }
    return false;
}

struct Flattener {
    SceneImpl &s;
    FlatScene &f;
    std::string err;
    // sub-BVHs are emitted after the world's nodes (the kernel stages nodes [0, n_world_nodes) in LDS)
    struct PendingSubBvh {
        uint32_t object;
        HostHittable tree;
        std::vector<uint32_t> ref_of;
    };
    std::vector<PendingSubBvh> pending;

    // When a BVH world mixes Sphere and MovingSphere leaves, static spheres are stored as moving-sphere rows with a
    // zero displacement (centre(t) = c0 + frac * 0 = c0 exactly), so that a wave's leaf tests run one code path
    // instead of two.  Not done if a centre component is -0.0 (c0 + 0.0 would flip it to +0.0).
    bool unify_spheres = false;
    const bool plain_quads = (s.options & RT_SCENE_PLAIN_QUADS) != 0;

    uint32_t add_primitive(const HostHittable &h, bool world_leaf = false)
    {
        if (h.kind == HKind::Sphere && unify_spheres && world_leaf) {
            f.mspheres.push_back({h.c0.x, h.c0.y, h.c0.z, 0.0, 0.0, 0.0, 0.0, 1.0, h.radius * h.radius});
            f.msphere_aux.push_back({1 / h.radius, h.material - 1, 0});
            return make_ref(REF_MSPHERE, (uint32_t)f.mspheres.size() - 1);
        }
        if (h.kind == HKind::Sphere) {
            f.spheres.push_back({h.c0.x, h.c0.y, h.c0.z, h.radius * h.radius});
            f.sphere_aux.push_back({1 / h.radius, h.material - 1, 0});
            return make_ref(REF_SPHERE, (uint32_t)f.spheres.size() - 1);
        }
        if (h.kind == HKind::MovingSphere) {
            D3 dc = sub(h.c1, h.c0);
            f.mspheres.push_back({h.c0.x, h.c0.y, h.c0.z, dc.x, dc.y, dc.z, h.t0, h.t1 - h.t0, h.radius * h.radius});
            f.msphere_aux.push_back({1 / h.radius, h.material - 1, 0});
            return make_ref(REF_MSPHERE, (uint32_t)f.mspheres.size() - 1);
        }
        f.quads.push_back({h.q.x, h.q.y, h.q.z, h.u.x, h.u.y, h.u.z, h.v.x, h.v.y, h.v.z, h.w.x, h.w.y, h.w.z,
                           h.normal.x, h.normal.y, h.normal.z, h.plane_d});
        // RT_SCENE_PLAIN_QUADS (tests): every quad takes the general test, boxes stay lists of six quads
        f.quad_aa.push_back(plain_quads ? AAQuad{} : axis_aligned(f.quads.back()));
        f.quad_mat.push_back(h.material - 1);
        return make_ref(REF_QUAD, (uint32_t)f.quads.size() - 1);
    }

    // Six consecutive quads that are exactly what MakeBox builds for some corners mn < mx (R/Instance.h:166-184: front,
    // right, back, left, top, bottom): every face lies in the plane of one corner coordinate, starts at a corner and
    // spans the rounded extent fl(mx - mn) towards the other one.  Anything else stays a plain list of quads.
    bool box_of_six(uint32_t first, BoxRec &b) const
    {
        static const uint32_t codes[6] = {1 + 3 * 2 + 0, 1 + 3 * 0 + 2, 1 + 3 * 2 + 0, 1 + 3 * 0 + 2, 1 + 3 * 1 + 0, 1 + 3 * 1 + 0};
        for (int k = 0; k < 6; k++)
            if (f.quad_aa[first + k].code != codes[k]) return false;
        const QuadGeom &front = f.quads[first + 0], &back = f.quads[first + 2], &top = f.quads[first + 4];
        const double mn[3] = {front.qx, front.qy, back.qz}, mx[3] = {back.qx, top.qy, front.qz};
        for (int k = 0; k < 3; k++)
            if (!(mn[k] < mx[k]) || !std::isfinite(mn[k]) || !std::isfinite(mx[k])) return false;
        for (int k = 0; k < 6; k++) {
            const QuadGeom &g = f.quads[first + k];
            const double q[3] = {g.qx, g.qy, g.qz}, u[3] = {g.ux, g.uy, g.uz}, v[3] = {g.vx, g.vy, g.vz};
            const int a = (int)(codes[k] - 1) / 3, p = (int)(codes[k] - 1) % 3, qa = 3 - a - p;
            if (q[a] != mn[a] && q[a] != mx[a]) return false;
            const int ax[2] = {p, qa};
            const double ev[2] = {u[p], v[qa]};
            for (int e = 0; e < 2; e++) {
                const double ext = mx[ax[e]] - mn[ax[e]];
                const bool from_min = q[ax[e]] == mn[ax[e]] && ev[e] == ext;
                const bool from_max = q[ax[e]] == mx[ax[e]] && ev[e] == -ext;
                if (!from_min && !from_max) return false;
            }
            b.na[k] = f.quad_aa[first + k].na;
            b.d[k] = f.quad_aa[first + k].d;
        }
        for (int k = 0; k < 3; k++) {
            b.mn[k] = mn[k];
            b.mx[k] = mx[k];
        }
        b.quad_first = first;
        return true;
    }

    // AAQuad of a quad whose edge vectors each lie along one coordinate axis (code 0 otherwise); see flat_scene.h.
    static AAQuad axis_aligned(const QuadGeom &g)
    {
        AAQuad out{};
        const double u[3] = {g.ux, g.uy, g.uz}, v[3] = {g.vx, g.vy, g.vz}, w[3] = {g.wx, g.wy, g.wz}, n[3] = {g.nx, g.ny, g.nz};
        const double q[3] = {g.qx, g.qy, g.qz};
        int p = -1, qa = -1;
        for (int k = 0; k < 3; k++) {
            if (u[k] != 0.0) p = p == -1 ? k : -2;
            if (v[k] != 0.0) qa = qa == -1 ? k : -2;
        }
        if (p < 0 || qa < 0 || p == qa) return out;
        const int a = 3 - p - qa;
        // everything the shortcut drops must be an exact zero, everything it keeps finite
        if (n[p] != 0.0 || n[qa] != 0.0 || w[p] != 0.0 || w[qa] != 0.0) return out;
        if (!std::isfinite(n[a]) || !std::isfinite(w[a]) || !std::isfinite(g.d) || n[a] == 0.0) return out;
        for (int k = 0; k < 3; k++)
            if (!std::isfinite(u[k]) || !std::isfinite(v[k]) || !std::isfinite(q[k])) return out;
        out.na = n[a];
        out.d = g.d;
        out.wa = w[a];
        out.qp = q[p];
        out.qq = q[qa];
        // component a of cross(ph, v) is ph[a+1]*v[a+2] - ph[a+2]*v[a+1]; of cross(u, ph): u[a+1]*ph[a+2] - u[a+2]*ph[a+1]
        const bool p_first = p == (a + 1) % 3;  // (p, q) = (a+1, a+2)
        out.kv = p_first ? v[qa] : -v[qa];
        out.ku = p_first ? u[p] : -u[p];
        out.code = 1u + 3u * (uint32_t)a + (uint32_t)p;
        return out;
    }

    static bool is_primitive(HKind k) { return k == HKind::Sphere || k == HKind::MovingSphere || k == HKind::Quad; }

    // Collect the primitives of a (possibly nested) list in visiting order.  A closest-hit scan over a
    // nested list equals the scan over its flattened sequence (R/HittableList.h:39-57 keeps one running
    // closestSoFar), provided the members draw no random numbers -- i.e. are primitives.
    bool collect_list(uint32_t handle, std::vector<uint32_t> &prims)
    {
        const HostHittable &h = s.hittables[handle - 1];
        if (is_primitive(h.kind)) {
            prims.push_back(handle);
            return true;
        }
        if (h.kind == HKind::List || h.kind == HKind::Bvh) {
            // A nested BvhNode over primitives returns the same closest hit as a scan of its leaves.
            for (uint32_t c : h.items)
                if (!collect_list(c, prims)) return false;
            return true;
        }
        err = "unsupported nesting: a list/BVH inside an instance or medium may only contain primitives and lists";
        return false;
    }

    // ---- general nesting: what ObjectRec cannot express stays a tree (flat_scene.h TreeNodeRec) ----
    bool only_primitives(uint32_t handle) const  // a (nested) list / BVH whose members are all primitives
    {
        const HostHittable &h = s.hittables[handle - 1];
        if (is_primitive(h.kind)) return true;
        if (h.kind != HKind::List && h.kind != HKind::Bvh) return false;
        for (uint32_t c : h.items)
            if (!only_primitives(c)) return false;
        return true;
    }
    // [ConstantMedium] -> Translate / RotateY chain -> a primitive or a list / BVH of primitives: the ObjectRec form
    bool fits_flat(uint32_t handle) const
    {
        const HostHittable *h = &s.hittables[handle - 1];
        if (h->kind == HKind::Medium) h = &s.hittables[h->child - 1];
        while (h->kind == HKind::Translate || h->kind == HKind::RotateY) h = &s.hittables[h->child - 1];
        if (h->kind == HKind::Medium) return false;
        return only_primitives((uint32_t)(h - s.hittables.data()) + 1);
    }
    uint32_t tree_depth_seen = 0;
    // One node of the tree for `handle`, whose Hit is called with the ray transformed by `chain` (outermost first).
    uint32_t lower_tree(uint32_t handle, const std::vector<Xform> &chain, uint32_t depth)
    {
        if (depth > tree_depth_seen) tree_depth_seen = depth;
        const HostHittable &h = s.hittables[handle - 1];
        TreeNodeRec rec{};
        rec.chain_first = (uint32_t)f.xforms.size();
        rec.chain_count = (uint32_t)chain.size();
        f.xforms.insert(f.xforms.end(), chain.begin(), chain.end());
        const uint32_t me = (uint32_t)f.tree_nodes.size();
        f.tree_nodes.push_back(rec);
        if (is_primitive(h.kind)) {
            rec.kind = TN_PRIM;
            rec.a = add_primitive(h);
        } else if (h.kind == HKind::Translate || h.kind == HKind::RotateY) {
            std::vector<Xform> inner = chain;
            if (h.kind == HKind::Translate) inner.push_back({h.offset.x, h.offset.y, h.offset.z, XF_TRANSLATE, 0});
            else inner.push_back({h.sin_t, h.cos_t, 0.0, XF_ROTATE_Y, 0});
            rec.kind = h.kind == HKind::Translate ? TN_TRANSLATE : TN_ROTATE_Y;
            rec.a = lower_tree(h.child, inner, depth + 1);
        } else if (h.kind == HKind::Medium) {
            f.media.push_back({h.neg_inv_density, h.material - 1, kNone, 0.0, 0.0, 0.0, 0.0});
            f.flags |= SCENE_HAS_MEDIA;
            rec.kind = TN_MEDIUM;
            rec.b = (uint32_t)f.media.size() - 1;
            rec.a = lower_tree(h.child, chain, depth + 1);
        } else if (h.kind == HKind::List) {
            rec.kind = TN_LIST;
            std::vector<uint32_t> kids;
            for (uint32_t c : h.items) kids.push_back(lower_tree(c, chain, depth + 1));
            rec.a = (uint32_t)f.tree_items.size();
            rec.b = (uint32_t)kids.size();
            f.tree_items.insert(f.tree_items.end(), kids.begin(), kids.end());
        } else {  // HKind::Bvh
            rec.kind = TN_BVH;
            std::vector<uint32_t> node_of(s.hittables.size() + 1, kNone);
            lower_bvh_leaves(h, chain, depth + 1, node_of);
            rec.a = thread_tree_general(h, 0, kNone, node_of);
        }
        f.tree_nodes[me] = rec;
        return me;
    }
    // Threaded nodes of a BvhNode inside a tree.  Unlike the world's (thread_tree), a child may be a BvhNode OBJECT: the
    // reference's loop cannot tell it from one of its own inner nodes (Hittable::IsBvhNode, R/BvhNode.h:124-143), so it is
    // spliced in as an inner child -- a node may then hold one leaf and one inner child, and a span-1 node over a BvhNode
    // object walks that object's tree twice.  a / b = REF_TREE | tree node for a leaf, the REF_INNER marker for an inner
    // child; the inner children follow the node in order, the first one at n + 1.
    // Leaves are lowered first (lower_bvh_leaves): a leaf may hold BvhNodes of its own, whose nodes must not land between a
    // node and its first inner child.
    void lower_bvh_leaves(const HostHittable &bvh, const std::vector<Xform> &chain, uint32_t depth, std::vector<uint32_t> &node_of)
    {
        for (const auto &tn : bvh.tree) {
            if (tn.left >= 0) continue;
            for (uint32_t hnd : {tn.leaf_a, tn.leaf_b}) {
                const HostHittable &obj = s.hittables[hnd - 1];
                if (obj.kind == HKind::Bvh) lower_bvh_leaves(obj, chain, depth, node_of);
                else if (node_of[hnd] == kNone) node_of[hnd] = lower_tree(hnd, chain, depth);
            }
        }
    }
    uint32_t thread_tree_general(const HostHittable &bvh, int ti, uint32_t escape, const std::vector<uint32_t> &node_of)
    {
        const auto &tn = bvh.tree[ti];
        const uint32_t me = (uint32_t)f.tree_bvh.size();
        f.tree_bvh.push_back({});
        BvhNodeRec rec{};
        rec.xlo = tn.box.lo[0]; rec.xhi = tn.box.hi[0];
        rec.ylo = tn.box.lo[1]; rec.yhi = tn.box.hi[1];
        rec.zlo = tn.box.lo[2]; rec.zhi = tn.box.hi[2];
        rec.escape = escape;
        // the two children, each an inner subtree (own tree node, or a BvhNode object) or a leaf
        struct Kid { bool inner; const HostHittable *owner; int index; uint32_t leaf; } kid[2];
        for (int c = 0; c < 2; c++) {
            if (tn.left >= 0) {
                kid[c] = {true, &bvh, c == 0 ? tn.left : tn.right, 0};
            } else {
                const uint32_t hnd = c == 0 ? tn.leaf_a : tn.leaf_b;
                const HostHittable &obj = s.hittables[hnd - 1];
                if (obj.kind == HKind::Bvh) kid[c] = {true, &obj, 0, 0};
                else kid[c] = {false, nullptr, 0, hnd};
            }
        }
        uint32_t refs[2];
        for (int c = 0; c < 2; c++)
            refs[c] = kid[c].inner ? make_ref(REF_INNER, 0) : make_ref(REF_TREE, node_of[kid[c].leaf]);
        rec.a = refs[0];
        rec.b = refs[1];
        f.tree_bvh[me] = rec;
        // inner children in order; the first one's walk escapes to the second one, the last one's to this node's escape
        const int n_inner = (kid[0].inner ? 1 : 0) + (kid[1].inner ? 1 : 0);
        int seen = 0;
        size_t first_begin = 0, first_end = 0;
        for (int c = 0; c < 2; c++) {
            if (!kid[c].inner) continue;
            seen++;
            if (seen == 1 && n_inner == 2) {
                first_begin = f.tree_bvh.size();
                thread_tree_general(*kid[c].owner, kid[c].index, kNone - 1 /* placeholder */, node_of);
                first_end = f.tree_bvh.size();
            } else {
                if (n_inner == 2)
                    for (size_t k = first_begin; k < first_end; k++)
                        if (f.tree_bvh[k].escape == kNone - 1) f.tree_bvh[k].escape = (uint32_t)f.tree_bvh.size();
                thread_tree_general(*kid[c].owner, kid[c].index, escape, node_of);
            }
        }
        return me;
    }

    // Lower one world leaf to a ref.
    uint32_t lower_leaf(uint32_t handle)
    {
        const HostHittable *h = &s.hittables[handle - 1];
        if (is_primitive(h->kind)) return add_primitive(*h, true);
        if (!fits_flat(handle)) {
            f.flags |= SCENE_HAS_TREES;
            return make_ref(REF_TREE, lower_tree(handle, {}, 1));
        }

        ObjectRec obj{};
        obj.medium = kNone;
        obj.coop_first = kNone;
        obj.coop_boxes = kNone;
        if (h->kind == HKind::Medium) {
            f.media.push_back({h->neg_inv_density, h->material - 1, kNone, 0.0, 0.0, 0.0, 0.0});
            obj.medium = (uint32_t)f.media.size() - 1;
            f.flags |= SCENE_HAS_MEDIA;
            h = &s.hittables[h->child - 1];
            if (h->kind == HKind::Medium) {
                err = "unsupported nesting: ConstantMedium directly inside ConstantMedium";
                return kNone;
            }
        }
        obj.xf_first = (uint32_t)f.xforms.size();
        while (h->kind == HKind::Translate || h->kind == HKind::RotateY) {
            if (h->kind == HKind::Translate)
                f.xforms.push_back({h->offset.x, h->offset.y, h->offset.z, XF_TRANSLATE, 0});
            else
                f.xforms.push_back({h->sin_t, h->cos_t, 0.0, XF_ROTATE_Y, 0});
            obj.xf_count++;
            h = &s.hittables[h->child - 1];
        }
        if (h->kind == HKind::Medium) {
            err = "unsupported nesting: ConstantMedium inside an instance transform";
            return kNone;
        }
        if (is_primitive(h->kind)) {
            obj.geom_kind = GEOM_SINGLE;
            obj.first = add_primitive(*h);
            obj.count = 1;
        } else {
            std::vector<uint32_t> prims;
            uint32_t self = (uint32_t)(h - s.hittables.data()) + 1;
            if (!collect_list(self, prims)) return kNone;
            bool all_s = true, all_m = true, all_q = true;
            for (uint32_t p : prims) {
                HKind k = s.hittables[p - 1].kind;
                all_s &= k == HKind::Sphere;
                all_m &= k == HKind::MovingSphere;
                all_q &= k == HKind::Quad;
            }
            obj.count = (uint32_t)prims.size();
            if (prims.size() >= kSubBvhMinPrims && !has_coincident_primitives(s, prims)) {
                // A closest-hit scan and a BVH over the same primitives return the same hit (the reference's own
                // invariant, Docs 2-3 BVH :733,:772); primitives draw no random numbers, so nothing else changes.
                HostHittable sub{};
                sub.kind = HKind::Bvh;
                std::vector<uint32_t> objs = prims;
                build_tree(&s, sub, objs, 0, (int)objs.size());
                std::vector<uint32_t> ref_of(s.hittables.size() + 1, kNone);
                const size_t spheres_before = f.spheres.size();
                for (uint32_t hnd : objs)
                    if (ref_of[hnd] == kNone) ref_of[hnd] = add_primitive(s.hittables[hnd - 1]);
                // all static spheres, each listed once: the rows just added are this group's, contiguously
                if (all_s && f.spheres.size() - spheres_before == prims.size()) {
                    obj.coop_first = (uint32_t)spheres_before;
                    obj.coop_boxes = (uint32_t)f.group_boxes.size();
                    for (size_t g0 = 0; g0 < prims.size(); g0 += kCoopGroup) {
                        GroupBox gb;
                        for (int k = 0; k < 3; k++) {
                            gb.lo[k] = DBL_MAX;
                            gb.hi[k] = -DBL_MAX;
                        }
                        for (size_t k = g0; k < g0 + kCoopGroup && k < prims.size(); k++) {
                            const SphereGeom &sg = f.spheres[spheres_before + k];
                            const double r = std::sqrt(sg.r2), c[3] = {sg.cx, sg.cy, sg.cz};
                            for (int a2 = 0; a2 < 3; a2++) {
                                // outwards by a part in 2^20 of the magnitudes involved: far beyond any rounding of the slab
                                // test or of the sphere's own arithmetic, far below anything that would cost culling power
                                const double pad = 9.5367431640625e-07 * (std::fabs(c[a2]) + r) + 1e-300;
                                gb.lo[a2] = std::fmin(gb.lo[a2], c[a2] - r - pad);
                                gb.hi[a2] = std::fmax(gb.hi[a2], c[a2] + r + pad);
                            }
                        }
                        f.group_boxes.push_back(gb);
                    }
                }
                obj.geom_kind = GEOM_BVH;
                obj.first = kNone;  // patched by emit_pending()
                pending.push_back({(uint32_t)f.objects.size(), std::move(sub), std::move(ref_of)});
            } else if (prims.empty()) {
                obj.geom_kind = GEOM_MIXED;
                obj.first = (uint32_t)f.items.size();
            } else if (all_s || all_m || all_q) {
                obj.geom_kind = all_s ? GEOM_SPHERES : (all_m ? GEOM_MSPHERES : GEOM_QUADS);
                uint32_t first_ref = add_primitive(s.hittables[prims[0] - 1]);
                obj.first = first_ref & kRefIndexMask;
                for (size_t k = 1; k < prims.size(); k++) add_primitive(s.hittables[prims[k] - 1]);
                if (all_q && prims.size() == 6) {
                    BoxRec b{};
                    if (box_of_six(obj.first, b)) {
                        obj.geom_kind = GEOM_BOX;
                        obj.first = (uint32_t)f.boxes.size();
                        f.boxes.push_back(b);
                    }
                }
            } else {
                obj.geom_kind = GEOM_MIXED;
                std::vector<uint32_t> refs;
                for (uint32_t p : prims) refs.push_back(add_primitive(s.hittables[p - 1]));
                obj.first = (uint32_t)f.items.size();
                f.items.insert(f.items.end(), refs.begin(), refs.end());
            }
        }
        // a plain box needs nothing from the object record: the leaf points at the box itself
        if (obj.geom_kind == GEOM_BOX && obj.xf_count == 0 && obj.medium == kNone) return make_ref(REF_BOX, obj.first);
        if (obj.medium != kNone && obj.geom_kind == GEOM_SINGLE && obj.xf_count == 0 && (obj.first >> kRefShift) == REF_SPHERE) {
            const SphereGeom &g = f.spheres[obj.first & kRefIndexMask];
            MediumRec &m = f.media[obj.medium];
            m.sphere = obj.first & kRefIndexMask;
            m.cx = g.cx; m.cy = g.cy; m.cz = g.cz; m.r2 = g.r2;
        }
        f.objects.push_back(obj);
        return make_ref(obj.medium != kNone ? REF_MOBJECT : REF_OBJECT, (uint32_t)f.objects.size() - 1);
    }

    void emit_pending()
    {
        for (auto &job : pending) f.objects[job.object].first = thread_tree(job.tree, 0, kNone, job.ref_of);
        pending.clear();
    }

    // Thread the reference's traversal (R/BvhNode.h:101-158) into escape links.  The reference visits
    // a node, tests leaf children on the spot, descends into the first inner child and stacks the
    // second; the stack therefore always holds the right siblings of the current root path, so "pop"
    // is a static successor: the escape link.
    uint32_t thread_tree(const HostHittable &bvh, int ti, uint32_t escape, const std::vector<uint32_t> &leaf_ref_of_handle)
    {
        const auto &tn = bvh.tree[ti];
        uint32_t me = (uint32_t)f.nodes.size();
        f.nodes.push_back({});
        BvhNodeRec rec{};
        rec.xlo = tn.box.lo[0]; rec.xhi = tn.box.hi[0];
        rec.ylo = tn.box.lo[1]; rec.yhi = tn.box.hi[1];
        rec.zlo = tn.box.lo[2]; rec.zhi = tn.box.hi[2];
        rec.escape = escape;
        if (tn.left < 0) {
            rec.a = leaf_ref_of_handle[tn.leaf_a];
            rec.b = leaf_ref_of_handle[tn.leaf_b];
            f.nodes[me] = rec;
            return me;
        }
        rec.a = rec.b = make_ref(REF_INNER, 0);
        f.nodes[me] = rec;
        // left subtree occupies [me+1, right_index); its escape is the right child
        // we do not know right_index until the left subtree is emitted: patch afterwards
        size_t left_begin = f.nodes.size();
        thread_tree(bvh, tn.left, kNone - 1 /* placeholder */, leaf_ref_of_handle);
        uint32_t right_index = (uint32_t)f.nodes.size();
        for (size_t k = left_begin; k < right_index; k++)
            if (f.nodes[k].escape == kNone - 1) f.nodes[k].escape = right_index;
        thread_tree(bvh, tn.right, escape, leaf_ref_of_handle);
        return me;
    }
};
} // namespace

static void lower_materials(SceneImpl &s, FlatScene &f)
{
    f.materials.clear();
    f.textures.clear();
    for (const HostTexture &t : s.textures) {
        TextureRec r{};
        r.kind = t.kind;
        r.r = t.color.x; r.g = t.color.y; r.b = t.color.z;
        r.s = t.s;
        if (t.kind == TEX_IMAGE || t.kind == TEX_NOISE) f.flags |= SCENE_RICH_TEXTURES;
        if (t.kind == TEX_CHECKER) {
            r.a = t.a - 1;
            r.b_ = t.b - 1;
        } else {
            r.a = t.a;
        }
        f.textures.push_back(r);
    }
    for (const HostMaterial &m : s.materials) {
        MaterialRec r{};
        r.kind = m.kind;
        r.tex = m.texture ? m.texture - 1 : kNone;
        r.r = m.albedo.x; r.g = m.albedo.y; r.b = m.albedo.z;
        r.p = m.p;
        if (m.texture) {
            const HostTexture &t = s.textures[m.texture - 1];
            auto put = [](double *dst, D3 c) { dst[0] = c.x; dst[1] = c.y; dst[2] = c.z; };
            if (t.kind == TEX_SOLID) {
                r.tex_inline = 1;
                put(r.even, t.color);
            } else if (t.kind == TEX_CHECKER && s.textures[t.a - 1].kind == TEX_SOLID && s.textures[t.b - 1].kind == TEX_SOLID) {
                r.tex_inline = 2;
                put(r.even, s.textures[t.a - 1].color);
                put(r.odd, s.textures[t.b - 1].color);
                r.inv_scale = t.s;
            } else if (t.kind == TEX_CHECKER) {
                f.flags |= SCENE_RICH_TEXTURES;  // nested checker: general kernel walks the table
            }
        }
        // U/V are only ever read by ImageTexture::Value; find out whether this material can reach one
        std::vector<uint32_t> todo;
        if (m.texture) todo.push_back(m.texture);
        while (!todo.empty()) {
            const HostTexture &t = s.textures[todo.back() - 1];
            todo.pop_back();
            if (t.kind == TEX_IMAGE) r.needs_uv = 1;
            if (t.kind == TEX_CHECKER) {
                todo.push_back(t.a);
                todo.push_back(t.b);
            }
        }
        f.materials.push_back(r);
    }
    f.images = s.images;
    f.image_bytes = s.image_bytes;
    f.perlin = s.perlin;
}

// ------------------------------------------------------------------------------------------------
// The library's own tree for primitive-only BVH worlds (flat_scene.h FastNodeRec): surface-area heuristic, full sweep
// over the three axes, bottom nodes of one or two primitives; then the eight octant threadings.
// ------------------------------------------------------------------------------------------------
namespace {
struct FastBuilder {
    const std::vector<Box> &boxes;       // the primitives' own boxes (the reference's, padded where thin)
    const std::vector<uint32_t> &refs;   // their leaf refs
    struct Node {
        Box box;
        int left = -1, right = -1;       // children, or -1
        uint32_t a = kNone, b = kNone;   // bottom node: one or two leaf refs
        int axis = 0;
    };
    std::vector<Node> nodes;

    static double area(const Box &b)
    {
        const double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
    int build(std::vector<uint32_t> &idx, size_t lo, size_t hi)
    {
        const int me = (int)nodes.size();
        nodes.push_back({});
        Node node;
        node.box = empty_box();
        for (size_t k = lo; k < hi; k++) node.box = box_merge(node.box, boxes[idx[k]]);
        const size_t n = hi - lo;
        if (n <= 2) {
            node.a = refs[idx[lo]];
            if (n == 2) node.b = refs[idx[lo + 1]];
            nodes[me] = node;
            return me;
        }
        // best split over the three axes: sort by box centre, sweep prefix / suffix areas
        double best_cost = DBL_MAX;
        int best_axis = 0;
        size_t best_at = lo + n / 2;
        std::vector<uint32_t> order(idx.begin() + lo, idx.begin() + hi), best_order;
        std::vector<double> suffix(n + 1);
        for (int axis = 0; axis < 3; axis++) {
            std::stable_sort(order.begin(), order.end(), [&](uint32_t p, uint32_t q) {
                return boxes[p].lo[axis] + boxes[p].hi[axis] < boxes[q].lo[axis] + boxes[q].hi[axis];
            });
            Box acc = empty_box();
            suffix[n] = 0.0;
            for (size_t k = n; k-- > 0;) {
                acc = box_merge(acc, boxes[order[k]]);
                suffix[k] = area(acc);
            }
            acc = empty_box();
            for (size_t k = 1; k < n; k++) {
                acc = box_merge(acc, boxes[order[k - 1]]);
                const double cost = area(acc) * (double)k + suffix[k] * (double)(n - k);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_at = lo + k;
                    best_order = order;
                }
            }
        }
        std::copy(best_order.begin(), best_order.end(), idx.begin() + lo);
        node.axis = best_axis;
        node.left = build(idx, lo, best_at);
        node.right = build(idx, best_at, hi);
        nodes[me] = node;
        return me;
    }
    // visiting order for direction octant `oct` (bit k set: direction component k is negative): the child on the side
    // the ray comes from first.  hit[n] = where to go when n's box is hit (inner nodes), esc[n] = where to go otherwise.
    void thread(int n, uint32_t escape, int oct, std::vector<uint16_t> &hit, std::vector<uint16_t> &esc) const
    {
        const Node &nd = nodes[n];
        esc[n] = (uint16_t)escape;
        if (nd.left < 0) {
            // a bottom node parks the lane (its leaves done, the walk goes to esc): the "hit" link says so -- bit 15, and in bits
            // 12-13 the kind of leaf a as the kind-batched kernels sort by it (render.hip leaf_kind) -- so that a node visit needs
            // no look at the leaf refs (kFastMaxNodes keeps node indices below bit 15)
            const uint32_t tag = nd.a >> kRefShift;
            const uint32_t kind = tag == REF_BOX ? 0u : (tag == REF_MOBJECT ? 1u : (tag == REF_OBJECT ? 2u : 3u));
            hit[n] = (uint16_t)(kFastBottom | (kind << 12));
            return;
        }
        const bool negative = (oct >> nd.axis) & 1;
        const int first = negative ? nd.right : nd.left, second = negative ? nd.left : nd.right;
        hit[n] = (uint16_t)first;
        thread(first, (uint32_t)second, oct, hit, esc);
        thread(second, escape, oct, hit, esc);
    }
};
} // namespace

static void build_fast_tree(FlatScene &f, bool reference_tree_only)
{
    f.fast_nodes.clear();
    if (reference_tree_only) return;  // RT_SCENE_REFERENCE_TREE_ONLY, or leaves that coincide (has_coincident_primitives)
    const size_t n = f.world_items.size();
    // (built for list worlds of primitives as well: RT_FLAG_ACCELERATE_LISTS renders them through it)
    if (n < 3 || n > 40000 || !f.objects.empty() || !f.boxes.empty() || !f.tree_nodes.empty()) return;
    for (uint32_t ref : f.world_items) {
        const uint32_t tag = ref >> kRefShift;
        if (tag != REF_SPHERE && tag != REF_MSPHERE && tag != REF_QUAD) return;
    }
    FastBuilder fb{f.leaf_boxes, f.world_items, {}};
    std::vector<uint32_t> idx(n);
    for (size_t k = 0; k < n; k++) idx[k] = (uint32_t)k;
    fb.build(idx, 0, n);
    if (fb.nodes.size() >= kFastMaxNodes) return;
    f.fast_nodes.resize(fb.nodes.size());
    for (size_t k = 0; k < fb.nodes.size(); k++) {
        const FastBuilder::Node &nd = fb.nodes[k];
        FastNodeRec &r = f.fast_nodes[k];
        r.xlo = nd.box.lo[0]; r.xhi = nd.box.hi[0];
        r.ylo = nd.box.lo[1]; r.yhi = nd.box.hi[1];
        r.zlo = nd.box.lo[2]; r.zhi = nd.box.hi[2];
        r.a = nd.left < 0 ? nd.a : make_ref(REF_INNER, 0);
        r.b = nd.left < 0 ? nd.b : make_ref(REF_INNER, 0);
    }
    std::vector<uint16_t> hit(fb.nodes.size()), esc(fb.nodes.size());
    for (int oct = 0; oct < 8; oct++) {
        fb.thread(0, kFastEnd, oct, hit, esc);
        for (size_t k = 0; k < fb.nodes.size(); k++) {
            f.fast_nodes[k].link[oct][0] = hit[k];
            f.fast_nodes[k].link[oct][1] = esc[k];
        }
    }
}

// The segmented walk of a BVH world with composite leaves (flat_scene.h FastOrder / SegMedium): the library's tree over the
// world's surface leaves, every node with the range of leaf positions below it, and the medium leaves in visiting order.
static void build_segment_tree(FlatScene &f, bool reference_tree_only)
{
    f.fast_order.clear();
    f.seg_media.clear();
    f.seg_cand.clear();
    if (reference_tree_only || f.world_kind != WORLD_BVH || !f.fast_nodes.empty()) return;
    const size_t n = f.world_items.size();
    if (n < 4 || n >= kSegEnd || (f.objects.empty() && f.boxes.empty())) return;
    std::vector<Box> boxes;
    std::vector<uint32_t> refs, order_of;
    std::vector<uint32_t> media;
    for (size_t k = 0; k < n; k++) {
        const uint32_t tag = f.world_items[k] >> kRefShift;
        if (tag == REF_TREE) return;  // general nesting: the interpreter's kernels
        if (tag == REF_MOBJECT) {
            media.push_back((uint32_t)k);
            continue;
        }
        boxes.push_back(f.leaf_boxes[k]);
        refs.push_back(f.world_items[k]);
        order_of.push_back((uint32_t)k);
    }
    if (media.size() > kSegMaxMedia || refs.size() < 3) return;
    FastBuilder fb{boxes, refs, {}};
    std::vector<uint32_t> idx(refs.size());
    for (size_t k = 0; k < idx.size(); k++) idx[k] = (uint32_t)k;
    fb.build(idx, 0, idx.size());
    if (fb.nodes.size() >= kFastMaxNodes) return;
    // position of a leaf ref in the world's list (refs are unique: every leaf was lowered once)
    std::unordered_map<uint32_t, uint32_t> position_of;
    for (size_t k = 0; k < refs.size(); k++) position_of[refs[k]] = order_of[k];
    auto position = [&](uint32_t ref) -> uint32_t { return position_of.at(ref); };
    f.fast_nodes.resize(fb.nodes.size());
    f.fast_order.resize(fb.nodes.size());
    for (size_t k = fb.nodes.size(); k-- > 0;) {  // children follow their parent in the array: bottom-up by going backwards
        const FastBuilder::Node &nd = fb.nodes[k];
        FastNodeRec &r = f.fast_nodes[k];
        r.xlo = nd.box.lo[0]; r.xhi = nd.box.hi[0];
        r.ylo = nd.box.lo[1]; r.yhi = nd.box.hi[1];
        r.zlo = nd.box.lo[2]; r.zhi = nd.box.hi[2];
        r.a = nd.left < 0 ? nd.a : make_ref(REF_INNER, 0);
        r.b = nd.left < 0 ? nd.b : make_ref(REF_INNER, 0);
        FastOrder &o = f.fast_order[k];
        if (nd.left < 0) {
            o.oa = (uint16_t)position(nd.a);
            o.ob = nd.b == kNone ? o.oa : (uint16_t)position(nd.b);
            o.omin = std::min(o.oa, o.ob);
            o.omax = std::max(o.oa, o.ob);
        } else {
            const FastOrder &l = f.fast_order[(size_t)nd.left], &rr = f.fast_order[(size_t)nd.right];
            o.oa = o.ob = 0;
            o.omin = std::min(l.omin, rr.omin);
            o.omax = std::max(l.omax, rr.omax);
        }
    }
    std::vector<uint16_t> hit(fb.nodes.size()), esc(fb.nodes.size());
    for (int oct = 0; oct < 8; oct++) {
        fb.thread(0, kFastEnd, oct, hit, esc);
        for (size_t k = 0; k < fb.nodes.size(); k++) {
            f.fast_nodes[k].link[oct][0] = hit[k];
            f.fast_nodes[k].link[oct][1] = esc[k];
        }
    }
    for (uint32_t k : media) {
        SegMedium m{};
        const Box &b = f.leaf_boxes[k];
        for (int a = 0; a < 3; a++) {
            // outwards by a part in 2^20 of the magnitudes involved (and at least the reference's own thin-box padding): far
            // beyond any rounding of the slab test or of the boundary's own arithmetic
            const double pad = 9.5367431640625e-07 * (std::fabs(b.lo[a]) + std::fabs(b.hi[a])) + 1e-4;
            m.lo[a] = b.lo[a] - pad;
            m.hi[a] = b.hi[a] + pad;
        }
        m.order = k;
        m.object = f.world_items[k] & kRefIndexMask;
        m.twice = 0;
        for (uint32_t nd = 0; nd < f.n_world_nodes; nd++)  // the reference's own tree: a span-1 node holds the leaf twice
            if (f.nodes[nd].a == f.world_items[k] && f.nodes[nd].b == f.world_items[k]) m.twice = 1;
        // the surface leaves that reach into the padded box (flat_scene.h SegMedium)
        m.cand_first = (uint32_t)f.seg_cand.size();
        m.cand_count = 0;
        for (size_t j = 0; j < refs.size() && m.cand_count != kNone; j++) {
            const Box &lb = boxes[j];
            bool meets = true;
            for (int a = 0; a < 3; a++) meets &= lb.lo[a] <= m.hi[a] && m.lo[a] <= lb.hi[a];
            if (!meets) continue;
            const uint32_t tag = refs[j] >> kRefShift;
            const bool simple = tag == REF_BOX || tag == REF_SPHERE || tag == REF_MSPHERE || tag == REF_QUAD;
            if (!simple || m.cand_count >= kSegMaxCandidates) {
                m.cand_count = kNone;
                f.seg_cand.resize(m.cand_first);
                break;
            }
            f.seg_cand.push_back({refs[j], order_of[j]});
            m.cand_count++;
        }
        f.seg_media.push_back(m);
    }
    f.flags |= SCENE_SEGMENTED;
}

int flatten_scene(SceneImpl &s)
{
    if (s.world == 0) return fail(RT_ERR_STATE, "rt_scene_commit: no world set (rt_scene_set_world)");
    if (!s.has_camera) return fail(RT_ERR_STATE, "rt_scene_commit: no camera set (rt_scene_set_camera)");
    if (s.launches_in_flight > 0)
        return fail(RT_ERR_STATE, "rt_scene_commit: a render of this scene is in flight (rt_render_finish it first)");
    s.committed = false;
    s.generation++;  // device copies made from the previous tables are stale from here on
    s.flat = FlatScene{};
    FlatScene &f = s.flat;
    lower_materials(s, f);
    Flattener fl{s, f, {}, {}};

    const HostHittable &world = s.hittables[s.world - 1];
    std::vector<uint32_t> leaves;  // hittable handles, final order
    if (world.kind == HKind::Bvh) {
        f.world_kind = WORLD_BVH;
        leaves = world.items;
    } else if (world.kind == HKind::List) {
        f.world_kind = WORLD_LIST;
        leaves = world.items;
    } else {
        f.world_kind = WORLD_LIST;  // a lone hittable as world behaves as a list of one
        leaves.push_back(s.world);
    }
    // A BvhNode OBJECT among a BvhNode world's leaves is walked by the reference as part of the world's own tree
    // (Hittable::IsBvhNode, R/BvhNode.h:124-143), not called as a leaf.  Over primitives only that makes no difference
    // (it becomes a group leaf); over composites the order of the leaf calls -- hence of the media's random draws -- does,
    // and the whole world is then kept as one tree (flat_scene.h TreeNodeRec, thread_tree_general).
    if (world.kind == HKind::Bvh) {
        bool nested_bvh = false;
        for (uint32_t h : leaves) nested_bvh |= s.hittables[h - 1].kind == HKind::Bvh && !fl.fits_flat(h);
        if (nested_bvh) {
            f.world_kind = WORLD_LIST;
            leaves.assign(1, s.world);
        }
    }
    if (f.world_kind == WORLD_BVH) {
        bool any_static = false, any_moving = false, negative_zero = false;
        for (uint32_t h : leaves) {
            const HostHittable &hh = s.hittables[h - 1];
            any_moving |= hh.kind == HKind::MovingSphere;
            if (hh.kind == HKind::Sphere) {
                any_static = true;
                negative_zero |= std::signbit(hh.c0.x) && hh.c0.x == 0.0;
                negative_zero |= std::signbit(hh.c0.y) && hh.c0.y == 0.0;
                negative_zero |= std::signbit(hh.c0.z) && hh.c0.z == 0.0;
            }
        }
        fl.unify_spheres = any_static && any_moving && !negative_zero;
    }
    std::vector<uint32_t> ref_of_handle(s.hittables.size() + 1, kNone);
    for (uint32_t h : leaves) {
        uint32_t ref = fl.lower_leaf(h);
        if (ref == kNone) return fail(RT_ERR_UNSUPPORTED, fl.err);
        ref_of_handle[h] = ref;
        f.world_items.push_back(ref);
        f.leaf_boxes.push_back(s.hittables[h - 1].box);
        const HKind hk = s.hittables[h - 1].kind;
        f.leaf_kinds.push_back(hk == HKind::Sphere ? 0 : (hk == HKind::MovingSphere ? 1 : (hk == HKind::Quad ? 2 : 3)));
    }
    if (f.world_kind == WORLD_BVH) {
        if (world.tree.empty()) return fail(RT_ERR_INVALID, "BvhNode world has no nodes");
        fl.thread_tree(world, 0, kNone, ref_of_handle);
        f.n_world_nodes = (uint32_t)f.nodes.size();
    } else {
        bool all_spheres = !f.world_items.empty();
        for (size_t k = 0; k < f.world_items.size(); k++) all_spheres &= f.world_items[k] == make_ref(REF_SPHERE, (uint32_t)k);
        if (all_spheres && f.spheres.size() == f.world_items.size()) f.flags |= SCENE_LIST_ALL_SPHERES;
    }
    fl.emit_pending();
    if (fl.tree_depth_seen > kTreeMaxDepth)
        return fail(RT_ERR_UNSUPPORTED, "object nesting deeper than " + std::to_string(kTreeMaxDepth) +
                                            " levels (the interpreter's stack; the reference's recursion is bounded by its 32 KiB stack)");
    {
        // What testing every leaf once costs, in half sphere tests (sphere 2, quad 3, box 12, a transform chain +4): the
        // launcher scans a small BVH world instead of walking it when this stays under a budget.  Measured: 16 spheres
        // scan 1.3x faster than they walk, 24 spheres and 8 free-standing boxes slower; the Cornell box (6 quads, 2
        // instanced boxes, rays that cross every node's box) 1.4x faster.
        uint64_t cost = 0;
        for (uint32_t ref : f.world_items) {
            const uint32_t tag = ref >> kRefShift, idx = ref & kRefIndexMask;
            if (tag == REF_SPHERE || tag == REF_MSPHERE) cost += 2;
            else if (tag == REF_QUAD) cost += 3;
            else if (tag == REF_BOX) cost += 12;
            else if (tag == REF_TREE) cost += 1000;
            else {
                const ObjectRec &o = f.objects[idx];
                cost += o.xf_count ? 4 : 0;
                if (o.geom_kind == GEOM_BOX) cost += 12;
                else if (o.geom_kind == GEOM_BVH) cost += 1000;
                else cost += 3ull * o.count;
            }
        }
        f.scan_cost = cost > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cost;
    }
    {
        const bool reference_only = (s.options & RT_SCENE_REFERENCE_TREE_ONLY) != 0 || has_coincident_primitives(s, leaves);
        build_fast_tree(f, reference_only);
        build_segment_tree(f, reference_only);
    }
    {
        // Rows of the list scan's conservative filter (render.hip filter_four): centre and |c|^2 - r^2, and the
        // scene's reach max(|c| + r) that bounds the filter's rounding error.  A non-finite row becomes (0, 0, 0, -inf):
        // such a sphere passes the filter for every ray and the exact test decides.
        f.sphere_scan.clear();
        f.scan_reach = 0.0;
        for (const SphereGeom &g : f.spheres) {
            const long double cc = (long double)g.cx * g.cx + (long double)g.cy * g.cy + (long double)g.cz * g.cz;
            double k = (double)(cc - (long double)g.r2);
            double reach = std::sqrt((double)cc) + std::sqrt(std::fabs(g.r2));
            if (!std::isfinite(k) || !std::isfinite(reach)) {
                f.sphere_scan.push_back(SphereScanRow{0.0, 0.0, 0.0, -std::numeric_limits<double>::infinity()});
                continue;
            }
            f.sphere_scan.push_back(SphereScanRow{g.cx, g.cy, g.cz, k});
            f.scan_reach = std::max(f.scan_reach, reach);
        }
        f.scan_reach *= 1.0 + 0x1p-40;  // the two square roots above were rounded
        // The packed fp32 form (flat_scene.h SphereScanPair).  Its margin grows with the square of the reach it has to cover, so
        // the few spheres that lie far outside the bulk -- the Book-1 ground sphere: |c| + r = 2000 in a scene of 15 -- are not
        // decided by it at all (k = -inf: always on to the exact test); the bulk is what stays within four times the median reach.
        {
            std::vector<double> reach(f.sphere_scan.size(), 0.0);
            for (size_t k = 0; k < f.sphere_scan.size(); k++) {
                const SphereScanRow &r = f.sphere_scan[k];
                const SphereGeom &g = f.spheres[k];
                reach[k] = std::isfinite(r.k) ? std::sqrt(r.cx * r.cx + r.cy * r.cy + r.cz * r.cz) + std::sqrt(std::fabs(g.r2)) : 0.0;
            }
            std::vector<double> sorted = reach;
            std::sort(sorted.begin(), sorted.end());
            const double bulk = sorted.empty() ? 0.0 : 4.0 * sorted[sorted.size() / 2];
            f.scan_reach32 = 0.0;
            f.sphere_scan32.assign((f.sphere_scan.size() + 1) / 2, SphereScanPair{});
            const float inf = std::numeric_limits<float>::infinity();
            for (size_t k = 0; k < 2 * f.sphere_scan32.size(); k++) {
                SphereScanPair &pr = f.sphere_scan32[k / 2];
                const int h = (int)(k & 1);
                if (k >= f.sphere_scan.size()) {  // padding: never passes
                    pr.cx[h] = pr.cy[h] = pr.cz[h] = 0.0f;
                    pr.k[h] = inf;
                    continue;
                }
                const SphereScanRow &r = f.sphere_scan[k];
                const bool decided = std::isfinite(r.k) && reach[k] <= bulk && reach[k] < 1e15;
                pr.cx[h] = decided ? (float)r.cx : 0.0f;
                pr.cy[h] = decided ? (float)r.cy : 0.0f;
                pr.cz[h] = decided ? (float)r.cz : 0.0f;
                pr.k[h] = decided ? (float)r.k : -inf;
                if (decided) f.scan_reach32 = std::max(f.scan_reach32, reach[k]);
            }
            f.scan_reach32 *= 1.0 + 0x1p-20;
        }
    }
    {
        bool unit_time = !f.mspheres.empty();
        for (const MSphereGeom &m : f.mspheres) unit_time &= (m.t0 == 0.0 && m.dt == 1.0);
        if (unit_time) f.flags |= SCENE_MS_UNIT_TIME;
        // A BVH world of nothing but spheres and unit-time moving spheres: thin waves scan all of them together instead of
        // walking (order-independent: no leaf draws random numbers).  The planes hold the leaves in leaf order, a static
        // sphere as a moving one that does not move (c0 + t * 0 = c0 exactly; not done when a centre component is -0.0).
        bool all_ms = f.world_kind == WORLD_BVH && !f.world_items.empty() && (f.mspheres.empty() || unit_time) && f.quads.empty() &&
                      f.objects.empty() && f.boxes.empty();
        for (size_t k = 0; all_ms && k < f.world_items.size(); k++) {
            const uint32_t tag = f.world_items[k] >> kRefShift, idx = f.world_items[k] & kRefIndexMask;
            all_ms = tag == REF_MSPHERE || tag == REF_SPHERE;
            if (tag == REF_SPHERE) {
                const SphereGeom &g = f.spheres[idx];
                all_ms = !(g.cx == 0.0 && std::signbit(g.cx)) && !(g.cy == 0.0 && std::signbit(g.cy)) && !(g.cz == 0.0 && std::signbit(g.cz));
            }
        }
        if (all_ms) {
            const size_t n = f.world_items.size(), np = (n + 63) & ~(size_t)63;
            f.ms_padded = (uint32_t)np;
            f.ms_planes.assign(7 * np, 0.0);
            for (size_t k = 0; k < np; k++) {
                const uint32_t ref = f.world_items[k < n ? k : 0];  // padding repeats leaf 0; the scan guards by index
                double row[7];
                if ((ref >> kRefShift) == REF_MSPHERE) {
                    const MSphereGeom &g = f.mspheres[ref & kRefIndexMask];
                    const double r7[7] = {g.c0x, g.c0y, g.c0z, g.dcx, g.dcy, g.dcz, g.r2};
                    std::memcpy(row, r7, sizeof row);
                } else {
                    const SphereGeom &g = f.spheres[ref & kRefIndexMask];
                    const double r7[7] = {g.cx, g.cy, g.cz, 0.0, 0.0, 0.0, g.r2};
                    std::memcpy(row, r7, sizeof row);
                }
                for (int p2 = 0; p2 < 7; p2++) f.ms_planes[(size_t)p2 * np + k] = row[p2];
            }
            f.flags |= SCENE_WORLD_MSPHERES;
        }
    }
    s.committed = true;
    return RT_OK;
}

SceneImpl::~SceneImpl()
{
    if (launches_in_flight > 0 || !films_in_flight.empty()) wait_for_films_in_flight(*this);  // kernels still read the tables
    for (DeviceTables *t : device)
        if (t) release_device_tables(t);
}

} // namespace rtow

// ================================================================================================
// C-ABI: construction
// ================================================================================================
using namespace rtow;

static inline SceneImpl *S(rt_scene *s) { return reinterpret_cast<SceneImpl *>(s); }

extern "C" {

const char *rt_last_error(void) { return g_error.c_str(); }
const char *rt_version(void) { return "rtow-hip 0.1 (gfx950)"; }

rt_rng *rt_rng_create_salted(uint64_t seed, uint64_t sequence, int salt_kind)
{
    RngImpl *r = new RngImpl;
    r->state = xorwow_seed(seed, salt_kind ? kSaltRocrand : kSaltCurandDevice);
    if (sequence) xorwow_skip_sequences(host_jump_table(), sequence, r->state);
    return reinterpret_cast<rt_rng *>(r);
}
rt_rng *rt_rng_create(uint64_t seed, uint64_t sequence) { return rt_rng_create_salted(seed, sequence, 0); }
void rt_rng_destroy(rt_rng *rng) { delete reinterpret_cast<RngImpl *>(rng); }
float rt_rng_uniform(rt_rng *rng) { return xorwow_uniform(reinterpret_cast<RngImpl *>(rng)->state); }
uint32_t rt_rng_next_u32(rt_rng *rng) { return xorwow_next(reinterpret_cast<RngImpl *>(rng)->state); }
void rt_rng_state(const rt_rng *rng, uint32_t out6[6])
{
    const Xorwow &s = reinterpret_cast<const RngImpl *>(rng)->state;
    out6[0] = s.d; out6[1] = s.v0; out6[2] = s.v1; out6[3] = s.v2; out6[4] = s.v3; out6[5] = s.v4;
}

rt_scene *rt_scene_create(void) { return reinterpret_cast<rt_scene *>(new SceneImpl); }
int rt_scene_set_options(rt_scene *scene, uint32_t options)
{
    if (!scene) return fail(RT_ERR_INVALID, "rt_scene_set_options: null scene");
    if (options & ~(uint32_t)(RT_SCENE_PLAIN_QUADS | RT_SCENE_REFERENCE_TREE_ONLY)) return fail(RT_ERR_INVALID, "rt_scene_set_options: unknown option bit");
    S(scene)->options = options;
    S(scene)->committed = false;  // takes effect with the next rt_scene_commit
    return RT_OK;
}
void rt_scene_destroy(rt_scene *scene) { delete S(scene); }

// ---- textures ----
rt_handle rt_solid_color(rt_scene *s, double r, double g, double b)
{
    if (!s) return 0;
    HostTexture t{};
    t.kind = TEX_SOLID;
    t.color = mk(r, g, b);
    return push_tex(S(s), t);
}
rt_handle rt_checker_texture(rt_scene *s, double scale, rt_handle even, rt_handle odd)
{
    if (!valid_tex(S(s), even) || !valid_tex(S(s), odd)) {
        set_error("rt_checker_texture: invalid texture handle");
        return 0;
    }
    HostTexture t{};
    t.kind = TEX_CHECKER;
    t.s = 1.0 / scale;  // Texture.h:64
    t.a = even;
    t.b = odd;
    return push_tex(S(s), t);
}
rt_handle rt_image_texture(rt_scene *s, const unsigned char *rgb, int width, int height)
{
    if (!s) return 0;
    SceneImpl *sc = S(s);
    ImageRec im{};
    im.offset = sc->image_bytes.size();
    if (rgb && width > 0 && height > 0) {
        im.width = width;
        im.height = height;
        sc->image_bytes.insert(sc->image_bytes.end(), rgb, rgb + (size_t)width * height * 3);
    } else {
        im.width = 0;
        im.height = 0;  // Texture.h:113-114: no data => cyan
    }
    sc->images.push_back(im);
    HostTexture t{};
    t.kind = TEX_IMAGE;
    t.a = (uint32_t)sc->images.size() - 1;
    return push_tex(sc, t);
}
void rt_rtwimage_bytes(const unsigned char *in, size_t count, unsigned char *out)
{
    // stbi__ldr_to_hdr: (float)(pow(byte / 255.0f, l2h_gamma = 2.2f) * l2h_scale = 1.0f)   [stb_image.h:1869]
    // RtwImage::FloatToByte: <= 0 -> 0, >= 1 -> 255, else (unsigned char)(256.0f * v)     [R/RtwImage.h:100-105]
    unsigned char lut[256];
    for (int b = 0; b < 256; b++) {
        float v = (float)(std::pow((double)((float)b / 255.0f), (double)2.2f) * (double)1.0f);
        lut[b] = v <= 0.0f ? 0 : (1.0f <= v ? 255 : (unsigned char)(256.0f * v));
    }
    for (size_t k = 0; k < count; k++) out[k] = lut[in[k]];
}

rt_handle rt_noise_texture(rt_scene *s, double scale, rt_rng *rng)
{
    if (!s || !rng) {
        set_error("rt_noise_texture: scene and rng are required");
        return 0;
    }
    SceneImpl *sc = S(s);
    Xorwow &st = reinterpret_cast<RngImpl *>(rng)->state;
    PerlinRec p{};
    // Perlin.h:27-35: 256 unit vectors from RandomVector(-1, 1) (arguments drawn left to right), then
    // three Fisher-Yates permutations (Perlin.h:105-116: int(uniform * (i + 1)) in fp32, clamped to i).
    for (int i = 0; i < 256; i++) {
        double range = 1.0 - (-1.0);
        double a = -1.0 + range * (double)xorwow_uniform(st);
        double b = -1.0 + range * (double)xorwow_uniform(st);
        double c = -1.0 + range * (double)xorwow_uniform(st);
        D3 u = normalize(mk(a, b, c));
        p.vec[i][0] = u.x; p.vec[i][1] = u.y; p.vec[i][2] = u.z;
    }
    int32_t *perms[3] = {p.perm_x, p.perm_y, p.perm_z};
    for (int t = 0; t < 3; t++) {
        int32_t *q = perms[t];
        for (int i = 0; i < 256; i++) q[i] = i;
        for (int i = 255; i > 0; i--) {
            int target = (int)(xorwow_uniform(st) * (float)(i + 1));
            if (target > i) target = i;
            int32_t tmp = q[i];
            q[i] = q[target];
            q[target] = tmp;
        }
    }
    sc->perlin.push_back(p);
    HostTexture t{};
    t.kind = TEX_NOISE;
    t.s = scale;
    t.a = (uint32_t)sc->perlin.size() - 1;
    return push_tex(sc, t);
}

// ---- materials ----
static rt_handle textured_material(rt_scene *s, uint32_t kind, rt_handle tex, const char *who)
{
    if (!valid_tex(S(s), tex)) {
        set_error(std::string(who) + ": invalid texture handle");
        return 0;
    }
    HostMaterial m{};
    m.kind = kind;
    m.texture = tex;
    return push_mat(S(s), m);
}
rt_handle rt_lambertian_tex(rt_scene *s, rt_handle texture) { return textured_material(s, MAT_LAMBERTIAN, texture, "rt_lambertian_tex"); }
rt_handle rt_lambertian(rt_scene *s, double r, double g, double b) { return s ? rt_lambertian_tex(s, rt_solid_color(s, r, g, b)) : 0; }
rt_handle rt_diffuse_light_tex(rt_scene *s, rt_handle texture) { return textured_material(s, MAT_DIFFUSE_LIGHT, texture, "rt_diffuse_light_tex"); }
rt_handle rt_diffuse_light(rt_scene *s, double r, double g, double b) { return s ? rt_diffuse_light_tex(s, rt_solid_color(s, r, g, b)) : 0; }
rt_handle rt_isotropic_tex(rt_scene *s, rt_handle texture) { return textured_material(s, MAT_ISOTROPIC, texture, "rt_isotropic_tex"); }
rt_handle rt_isotropic(rt_scene *s, double r, double g, double b) { return s ? rt_isotropic_tex(s, rt_solid_color(s, r, g, b)) : 0; }
rt_handle rt_metal(rt_scene *s, double r, double g, double b, double fuzz)
{
    if (!s) return 0;
    HostMaterial m{};
    m.kind = MAT_METAL;
    m.albedo = mk(r, g, b);
    m.p = fuzz < 1.0 ? fuzz : 1.0;  // Metal.h:14
    return push_mat(S(s), m);
}
rt_handle rt_dielectric(rt_scene *s, double refraction_index)
{
    if (!s) return 0;
    HostMaterial m{};
    m.kind = MAT_DIELECTRIC;
    m.p = refraction_index;
    return push_mat(S(s), m);
}

// ---- hittables ----
rt_handle rt_sphere(rt_scene *s, double cx, double cy, double cz, double radius, rt_handle material)
{
    if (!valid_mat(S(s), material)) {
        set_error("rt_sphere: invalid material handle");
        return 0;
    }
    HostHittable h{};
    h.kind = HKind::Sphere;
    h.c0 = mk(cx, cy, cz);
    h.radius = radius;
    h.material = material;
    D3 rv = mk(radius, radius, radius);
    h.box = box_from_corners(sub(h.c0, rv), add(h.c0, rv));  // Sphere.h:18-19
    return push_h(S(s), std::move(h));
}
rt_handle rt_moving_sphere(rt_scene *s, double c0x, double c0y, double c0z, double c1x, double c1y, double c1z,
                           double time0, double time1, double radius, rt_handle material)
{
    if (!valid_mat(S(s), material)) {
        set_error("rt_moving_sphere: invalid material handle");
        return 0;
    }
    HostHittable h{};
    h.kind = HKind::MovingSphere;
    h.c0 = mk(c0x, c0y, c0z);
    h.c1 = mk(c1x, c1y, c1z);
    h.t0 = time0;
    h.t1 = time1;
    h.radius = radius;
    h.material = material;
    D3 rv = mk(radius, radius, radius);
    Box b0 = box_from_corners(sub(h.c0, rv), add(h.c0, rv));
    Box b1 = box_from_corners(sub(h.c1, rv), add(h.c1, rv));
    h.box = box_merge(b0, b1);  // MovingSphere.h:32-35
    return push_h(S(s), std::move(h));
}
rt_handle rt_quad(rt_scene *s, const double q[3], const double u[3], const double v[3], rt_handle material)
{
    if (!valid_mat(S(s), material) || !q || !u || !v) {
        set_error("rt_quad: invalid material handle or null vector");
        return 0;
    }
    HostHittable h{};
    h.kind = HKind::Quad;
    h.q = mk(q[0], q[1], q[2]);
    h.u = mk(u[0], u[1], u[2]);
    h.v = mk(v[0], v[1], v[2]);
    h.material = material;
    D3 n = cross(h.u, h.v);  // Quad.h:33-37
    h.normal = normalize(n);
    h.plane_d = dot(h.normal, h.q);
    h.w = over(n, dot(n, n));
    Box d1 = box_from_corners(h.q, add(add(h.q, h.u), h.v));  // Quad.h:44-49
    Box d2 = box_from_corners(add(h.q, h.u), add(h.q, h.v));
    h.box = box_merge(d1, d2);
    return push_h(S(s), std::move(h));
}
rt_handle rt_translate(rt_scene *s, rt_handle object, double ox, double oy, double oz)
{
    HostHittable *c = get_h(S(s), object);
    if (!c) {
        set_error("rt_translate: invalid object handle");
        return 0;
    }
    HostHittable h{};
    h.kind = HKind::Translate;
    h.child = object;
    h.offset = mk(ox, oy, oz);
    h.box = box_moved(c->box, h.offset);  // Instance.h:36
    return push_h(S(s), std::move(h));
}
rt_handle rt_rotate_y(rt_scene *s, rt_handle object, double angle_degrees)
{
    HostHittable *c = get_h(S(s), object);
    if (!c) {
        set_error("rt_rotate_y: invalid object handle");
        return 0;
    }
    HostHittable h{};
    h.kind = HKind::RotateY;
    h.child = object;
    double radians = angle_degrees * 3.1415926535897932385 / 180.0;  // Instance.h:78-80
    h.sin_t = std::sin(radians);
    h.cos_t = std::cos(radians);
    const Box cb = c->box;
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int i = 0; i < 2; i++)  // Instance.h:87-109: rotate the 8 corners
        for (int j = 0; j < 2; j++)
            for (int k = 0; k < 2; k++) {
                double x = i * cb.hi[0] + (1 - i) * cb.lo[0];
                double y = j * cb.hi[1] + (1 - j) * cb.lo[1];
                double z = k * cb.hi[2] + (1 - k) * cb.lo[2];
                double nx = h.cos_t * x + h.sin_t * z;
                double nz = -h.sin_t * x + h.cos_t * z;
                double t[3] = {nx, y, nz};
                for (int c2 = 0; c2 < 3; c2++) {
                    mn[c2] = std::fmin(mn[c2], t[c2]);
                    mx[c2] = std::fmax(mx[c2], t[c2]);
                }
            }
    h.box = box_from_corners(mk(mn[0], mn[1], mn[2]), mk(mx[0], mx[1], mx[2]));
    return push_h(S(s), std::move(h));
}
rt_handle rt_hittable_list(rt_scene *s, const rt_handle *objects, int count)
{
    if (!s || count < 0 || (count > 0 && !objects)) {
        set_error("rt_hittable_list: bad arguments");
        return 0;
    }
    HostHittable h{};
    h.kind = HKind::List;
    h.box = empty_box();
    for (int i = 0; i < count; i++) {
        HostHittable *c = get_h(S(s), objects[i]);
        if (!c) {
            set_error("rt_hittable_list: invalid child handle");
            return 0;
        }
        h.box = box_merge(h.box, c->box);  // HittableList.h:25-26
        h.items.push_back(objects[i]);
    }
    return push_h(S(s), std::move(h));
}
rt_handle rt_make_box(rt_scene *s, const double a[3], const double b[3], rt_handle material)
{
    if (!valid_mat(S(s), material) || !a || !b) {
        set_error("rt_make_box: invalid material handle or null corner");
        return 0;
    }
    // Instance.h:166-184: six quads, in the reference's order and with its (negated) edge vectors.
    double mn[3], mx[3];
    for (int k = 0; k < 3; k++) {
        mn[k] = std::fmin(a[k], b[k]);
        mx[k] = std::fmax(a[k], b[k]);
    }
    D3 dx = mk(mx[0] - mn[0], 0, 0), dy = mk(0, mx[1] - mn[1], 0), dz = mk(0, 0, mx[2] - mn[2]);
    D3 ndx = neg(dx), ndz = neg(dz);
    struct Side { double q[3]; D3 u, v; } sides[6] = {
        {{mn[0], mn[1], mx[2]}, dx, dy},    // front
        {{mx[0], mn[1], mx[2]}, ndz, dy},   // right
        {{mx[0], mn[1], mn[2]}, ndx, dy},   // back
        {{mn[0], mn[1], mn[2]}, dz, dy},    // left
        {{mn[0], mx[1], mx[2]}, dx, ndz},   // top
        {{mn[0], mn[1], mn[2]}, dx, dz},    // bottom
    };
    rt_handle quads[6];
    for (int k = 0; k < 6; k++) {
        double u[3] = {sides[k].u.x, sides[k].u.y, sides[k].u.z}, v[3] = {sides[k].v.x, sides[k].v.y, sides[k].v.z};
        quads[k] = rt_quad(s, sides[k].q, u, v, material);
        if (!quads[k]) return 0;
    }
    return rt_hittable_list(s, quads, 6);
}
static rt_handle medium_common(rt_scene *s, rt_handle boundary, double density, rt_handle phase)
{
    HostHittable *c = get_h(S(s), boundary);
    if (!c || !phase) {
        set_error("rt_constant_medium: invalid boundary or texture handle");
        return 0;
    }
    HostHittable h{};
    h.kind = HKind::Medium;
    h.child = boundary;
    h.neg_inv_density = -1.0 / density;  // ConstantMedium.h:41
    h.material = phase;
    h.box = c->box;  // ConstantMedium.h:96
    return push_h(S(s), std::move(h));
}
rt_handle rt_constant_medium(rt_scene *s, rt_handle boundary, double density, double r, double g, double b)
{
    if (!s) return 0;
    return medium_common(s, boundary, density, rt_isotropic(s, r, g, b));
}
rt_handle rt_constant_medium_tex(rt_scene *s, rt_handle boundary, double density, rt_handle texture)
{
    if (!s) return 0;
    return medium_common(s, boundary, density, rt_isotropic_tex(s, texture));
}
rt_handle rt_bvh_node(rt_scene *s, rt_handle *objects, int count)
{
    if (!s || !objects || count <= 0) {
        set_error("rt_bvh_node: needs at least one object (the reference's constructor does not terminate on an empty span)");
        return 0;
    }
    SceneImpl *sc = S(s);
    for (int i = 0; i < count; i++)
        if (!get_h(sc, objects[i])) {
            set_error("rt_bvh_node: invalid object handle");
            return 0;
        }
    HostHittable h{};
    h.kind = HKind::Bvh;
    std::vector<uint32_t> objs(objects, objects + count);
    build_tree(sc, h, objs, 0, count);
    h.box = h.tree[0].box;
    h.items = objs;
    std::memcpy(objects, objs.data(), sizeof(rt_handle) * (size_t)count);  // the reference sorts list[] in place
    return push_h(sc, std::move(h));
}
int rt_hittable_bounding_box(rt_scene *s, rt_handle object, double out[6])
{
    HostHittable *h = get_h(S(s), object);
    if (!h || !out) return fail(RT_ERR_INVALID, "rt_hittable_bounding_box: invalid handle");
    for (int k = 0; k < 3; k++) {
        out[2 * k] = h->box.lo[k];
        out[2 * k + 1] = h->box.hi[k];
    }
    return RT_OK;
}

int rt_scene_set_world(rt_scene *s, rt_handle world)
{
    if (!get_h(S(s), world)) return fail(RT_ERR_INVALID, "rt_scene_set_world: invalid handle");
    S(s)->world = world;
    S(s)->committed = false;
    return RT_OK;
}

int rt_scene_set_camera(rt_scene *s, const double lookfrom[3], const double lookat[3], const double vup[3],
                        double vfov_degrees, double aspect, double aperture, double focus_dist, double time0,
                        double time1, const double background[3])
{
    if (!s || !lookfrom || !lookat || !vup || !background) return fail(RT_ERR_INVALID, "rt_scene_set_camera: null argument");
    if (S(s)->launches_in_flight > 0)
        return fail(RT_ERR_STATE, "rt_scene_set_camera: a render of this scene is in flight (rt_render_finish it first)");
    // Camera.h:47-72
    CameraRec c{};
    D3 from = mk(lookfrom[0], lookfrom[1], lookfrom[2]), at = mk(lookat[0], lookat[1], lookat[2]);
    D3 up = mk(vup[0], vup[1], vup[2]);
    double theta = vfov_degrees * 3.14159265358979323846 / 180.0;
    double half_h = std::tan(theta / 2.0);
    double half_w = aspect * half_h;
    D3 w = normalize(sub(from, at));
    D3 u = normalize(cross(up, w));
    D3 v = cross(w, u);
    D3 llc = sub(sub(sub(from, scale(half_w * focus_dist, u)), scale(half_h * focus_dist, v)), scale(focus_dist, w));
    D3 horiz = scale(2.0 * half_w * focus_dist, u);
    D3 vert = scale(2.0 * half_h * focus_dist, v);
    auto put = [](double *dst, D3 a) { dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; };
    put(c.bg, mk(background[0], background[1], background[2]));
    put(c.origin, from);
    put(c.llc, llc);
    put(c.horizontal, horiz);
    put(c.vertical, vert);
    put(c.u, u);
    put(c.v, v);
    put(c.w, w);
    c.lens_radius = aperture / 2.0;
    c.time0 = time0;
    c.time1 = time1;
    S(s)->camera = c;
    S(s)->has_camera = true;
    S(s)->generation++;  // the uploaded CameraRec is stale; the flat tables do not depend on the camera
    return RT_OK;
}

int rt_scene_commit(rt_scene *s)
{
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_commit: null scene");
    return flatten_scene(*S(s));
}

int rt_scene_get_info(rt_scene *s, rt_scene_info *out)
{
    if (!s || !out) return fail(RT_ERR_INVALID, "rt_scene_get_info: null argument");
    if (!S(s)->committed) return fail(RT_ERR_STATE, "rt_scene_get_info: scene not committed");
    const FlatScene &f = S(s)->flat;
    std::memset(out, 0, sizeof *out);
    out->world_kind = f.world_kind;
    out->n_leaves = (uint32_t)f.world_items.size();
    out->n_nodes = (uint32_t)f.nodes.size();
    out->n_spheres = (uint32_t)f.spheres.size();
    out->n_moving_spheres = (uint32_t)f.mspheres.size();
    out->n_quads = (uint32_t)f.quads.size();
    out->n_objects = (uint32_t)f.objects.size();
    out->n_xforms = (uint32_t)f.xforms.size();
    out->n_media = (uint32_t)f.media.size();
    out->n_materials = (uint32_t)f.materials.size();
    out->n_textures = (uint32_t)f.textures.size();
    out->n_perlin = (uint32_t)f.perlin.size();
    out->n_images = (uint32_t)f.images.size();
    out->table_bytes = (uint32_t)(f.nodes.size() * sizeof(BvhNodeRec) + f.spheres.size() * sizeof(SphereGeom) +
                                  f.mspheres.size() * sizeof(MSphereGeom) + f.quads.size() * sizeof(QuadGeom) +
                                  f.objects.size() * sizeof(ObjectRec) + f.xforms.size() * sizeof(Xform) +
                                  f.materials.size() * sizeof(MaterialRec));
    out->image_bytes = (uint32_t)f.image_bytes.size();
    return RT_OK;
}

int rt_scene_dump_leaves(rt_scene *s, int max_leaves, int *kind_out, double *box_out)
{
    if (!s || !S(s)->committed) return -fail(RT_ERR_STATE, "rt_scene_dump_leaves: scene not committed");
    const FlatScene &f = S(s)->flat;
    int n = (int)f.world_items.size();
    for (int k = 0; k < n && k < max_leaves; k++) {
        if (kind_out) kind_out[k] = f.leaf_kinds[k];
        if (box_out)
            for (int a = 0; a < 3; a++) {
                box_out[6 * k + 2 * a] = f.leaf_boxes[k].lo[a];
                box_out[6 * k + 2 * a + 1] = f.leaf_boxes[k].hi[a];
            }
    }
    return n;
}
int rt_scene_dump_nodes(rt_scene *s, int max_nodes, double *box_out, uint32_t *abe_out)
{
    if (!s || !S(s)->committed) return -fail(RT_ERR_STATE, "rt_scene_dump_nodes: scene not committed");
    const FlatScene &f = S(s)->flat;
    int n = (int)f.nodes.size();
    for (int k = 0; k < n && k < max_nodes; k++) {
        const BvhNodeRec &r = f.nodes[k];
        if (box_out) {
            double *b = box_out + 6 * k;
            b[0] = r.xlo; b[1] = r.xhi; b[2] = r.ylo; b[3] = r.yhi; b[4] = r.zlo; b[5] = r.zhi;
        }
        if (abe_out) {
            abe_out[3 * k] = r.a;
            abe_out[3 * k + 1] = r.b;
            abe_out[3 * k + 2] = r.escape;
        }
    }
    return n;
}
int rt_scene_dump_fast_nodes(rt_scene *s, int max_nodes, double *box_out, uint32_t *ab_out, uint16_t *link_out)
{
    if (!s || !S(s)->committed) return -fail(RT_ERR_STATE, "rt_scene_dump_fast_nodes: scene not committed");
    const FlatScene &f = S(s)->flat;
    int n = (int)f.fast_nodes.size();
    for (int k = 0; k < n && k < max_nodes; k++) {
        const FastNodeRec &r = f.fast_nodes[k];
        if (box_out) {
            double *b = box_out + 6 * k;
            b[0] = r.xlo; b[1] = r.xhi; b[2] = r.ylo; b[3] = r.yhi; b[4] = r.zlo; b[5] = r.zhi;
        }
        if (ab_out) {
            ab_out[2 * k] = r.a;
            ab_out[2 * k + 1] = r.b;
        }
        if (link_out) std::memcpy(link_out + 16 * k, r.link, sizeof r.link);
    }
    return n;
}
int rt_scene_dump_camera(rt_scene *s, double out27[27])
{
    if (!s || !S(s)->has_camera || !out27) return fail(RT_ERR_STATE, "rt_scene_dump_camera: no camera");
    const CameraRec &c = S(s)->camera;
    const double *vs[8] = {c.bg, c.origin, c.llc, c.horizontal, c.vertical, c.u, c.v, c.w};
    for (int k = 0; k < 8; k++) std::memcpy(out27 + 3 * k, vs[k], 3 * sizeof(double));
    out27[24] = c.lens_radius;
    out27[25] = c.time0;
    out27[26] = c.time1;
    return RT_OK;
}

int rt_stripe_rows(int height, int stripe_rows, int rank, int world_size, int *rows_out, int max_rows)
{
    if (height < 0 || stripe_rows <= 0 || world_size <= 0 || rank < 0 || rank >= world_size) return -1;
    int n = 0;
    for (int j = 0; j < height; j++)
        if ((j / stripe_rows) % world_size == rank) {
            if (rows_out && n < max_rows) rows_out[n] = j;
            n++;
        }
    return n;
}

int rt_deinterleave(const double *gathered, int width, int height, int stripe_rows, int world_size,
                    size_t rank_stride_doubles, double *frame_full)
{
    if (!gathered || !frame_full || width <= 0 || height <= 0 || stripe_rows <= 0 || world_size <= 0)
        return fail(RT_ERR_INVALID, "rt_deinterleave: bad arguments");
    std::vector<size_t> next(world_size, 0);
    for (int j = 0; j < height; j++) {
        int r = (j / stripe_rows) % world_size;
        const double *src = gathered + (size_t)r * rank_stride_doubles + next[r] * (size_t)width * 3;
        std::memcpy(frame_full + (size_t)j * width * 3, src, sizeof(double) * (size_t)width * 3);
        next[r]++;
    }
    return RT_OK;
}

int rt_write_ppm_binary(const char *path, const double *frame, int width, int height)
{
    if (!path || !frame || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "rt_write_ppm_binary: bad arguments");
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return fail(RT_ERR_INVALID, std::string("rt_write_ppm_binary: cannot open ") + path);
    std::fprintf(fp, "P6\n%d %d\n255\n", width, height);
    std::vector<unsigned char> row((size_t)width * 3);
    for (int j = height - 1; j >= 0; j--) {  // same clamp and int(256 * c) as R/kernel.cu:710-718
        for (int i = 0; i < width * 3; i++) {
            double c = frame[(size_t)j * width * 3 + i];
            double x = c < 0.0 ? 0.0 : (c > 0.999 ? 0.999 : c);
            row[i] = (unsigned char)(int)(256.0 * x);
        }
        std::fwrite(row.data(), 1, row.size(), fp);
    }
    std::fclose(fp);
    return RT_OK;
}

int rt_write_pfm(const char *path, const double *frame, int width, int height)
{
    if (!path || !frame || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "rt_write_pfm: bad arguments");
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return fail(RT_ERR_INVALID, std::string("rt_write_pfm: cannot open ") + path);
    std::fprintf(fp, "PF\n%d %d\n-1.0\n", width, height);
    std::vector<float> row((size_t)width * 3);
    for (int j = 0; j < height; j++) {  // PFM stores the bottom row first: the frame's own order (j = 0 bottom)
        for (int i = 0; i < width * 3; i++) row[i] = (float)frame[(size_t)j * width * 3 + i];
        std::fwrite(row.data(), sizeof(float), row.size(), fp);
    }
    std::fclose(fp);
    return RT_OK;
}

int rt_write_ppm(const char *path, const double *frame, int width, int height)
{
    if (!path || !frame || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "rt_write_ppm: bad arguments");
    FILE *fp = std::fopen(path, "w");
    if (!fp) return fail(RT_ERR_INVALID, std::string("rt_write_ppm: cannot open ") + path);
    // R/kernel.cu:696-721: text P3, top row (j = H-1) first, clamp to [0, 0.999], int(256 * c)
    std::string buf;
    buf.reserve((size_t)width * 12 * 64);
    std::fprintf(fp, "P3\n%d %d\n255\n", width, height);
    char line[64];
    for (int j = height - 1; j >= 0; j--) {
        buf.clear();
        for (int i = 0; i < width; i++) {
            const double *c = frame + ((size_t)j * width + i) * 3;
            int q[3];
            for (int k = 0; k < 3; k++) {
                double x = c[k] < 0.0 ? 0.0 : (c[k] > 0.999 ? 0.999 : c[k]);
                q[k] = (int)(256.0 * x);
            }
            int n = std::snprintf(line, sizeof line, "%d %d %d\n", q[0], q[1], q[2]);
            buf.append(line, (size_t)n);
        }
        std::fwrite(buf.data(), 1, buf.size(), fp);
    }
    std::fclose(fp);
    return RT_OK;
}

} // extern "C"
