// flat_scene.h -- the SoA scene tables the HIP megakernel reads.  Written by the host flattener
// (scene_builder.cpp), uploaded once per device (device_scene.cpp), never touched by the oracle.
//
// Everything the reference keeps as heap objects with vptrs (R/Hittable.h:33-65, R/Material.h:27-44,
// R/Texture.h:24-30) becomes a tagged row in one of these tables; virtual dispatch becomes a switch on
// the tag; the pointer BVH (R/BvhNode.h:165-167) becomes a preorder array with escape links.
#pragma once
#include <stdint.h>

namespace rtow {

// ---- references to things that can be hit: (tag << 28) | index ----
enum : uint32_t {
    REF_SPHERE = 0u,   // index into spheres
    REF_MSPHERE = 1u,  // index into moving spheres
    REF_QUAD = 2u,     // index into quads
    REF_OBJECT = 3u,   // index into objects (instances / boxes / lists / media)
    REF_MEDIUM = 4u,   // only in hit results: index into media
    REF_BOX = 5u,      // leaves only: index into boxes (a MakeBox box without transform or medium; its hits are REF_QUAD)
    REF_MOBJECT = 6u,  // leaves only: index into objects, for an object that is a ConstantMedium (its test draws random numbers)
    REF_TREE = 7u,     // index into tree_nodes: a composite that does not fit ObjectRec, kept as the reference's own object tree
    REF_INNER = 14u,   // BVH node marker: children are the next node and the escape target
    REF_NONE = 15u
};
constexpr uint32_t kRefShift = 28;
constexpr uint32_t kRefIndexMask = (1u << kRefShift) - 1u;
constexpr uint32_t kNone = 0xFFFFFFFFu;
static inline constexpr uint32_t make_ref(uint32_t tag, uint32_t index) { return (tag << kRefShift) | index; }

// Sphere (R/Sphere.h:12-20).  geom = what the discriminant test needs; aux = what shading needs.
struct SphereGeom { double cx, cy, cz, r2; };            // r2 = radius * radius
// The same sphere as the conservative filter of the list scan reads it (render.hip filter_four): k = |c|^2 - r^2.
struct SphereScanRow { double cx, cy, cz, k; };
// Two spheres of the list as the PACKED fp32 form of that filter reads them (render.hip filter_pairs): gfx950 issues a
// v_pk_fma_f32 -- two fp32 fmas per lane -- in the slot of one fp64 fma, so a conservative fp32 filter over pairs of spheres
// costs 9 instructions per TWO spheres where the fp64 one costs 8 per sphere.  k = fl32(|c|^2 - r^2); -inf = "always passes"
// (a sphere beyond the reach the fp32 margin is sized for, e.g. the ground sphere of radius 1000, or a non-finite row);
// +inf = padding of an odd count.
struct SphereScanPair { float cx[2], cy[2], cz[2], k[2]; };
struct SphereAux { double inv_r; uint32_t mat; uint32_t pad; };

// MovingSphere (R/MovingSphere.h:19-36): centre(t) = c0 + ((t - t0) / dt) * dc
struct MSphereGeom { double c0x, c0y, c0z, dcx, dcy, dcz, t0, dt, r2; };

// Quad (R/Quad.h:25-49): plane n.x = d, w = n / (n.n) for the planar coordinates.
struct QuadGeom { double qx, qy, qz, ux, uy, uz, vx, vy, vz, wx, wy, wz, nx, ny, nz, d; };
// The same quad when u and v each lie along one coordinate axis (every quad MakeBox and the reference's scenes
// build): normal and w then have a single non-zero component, every other term of R/Quad.h:52-99's dot and cross
// products is an exact zero, and t / alpha / beta come out of one multiply each -- the same bits as the general
// test.  a = axis of the normal, p = axis of u, q = axis of v:
//   t = (d - na*o[a]) / (na*dir[a]);  ph = (o + t*dir) - Q;  alpha = wa*(ph[p]*kv);  beta = wa*(ku*ph[q])
// with kv = +-v[q], ku = +-u[p] (the sign of that term in the cross product).
struct AAQuad { double na, d, wa, qp, qq, ku, kv; uint32_t code, pad; };  // code 0: not axis-aligned; else 1 + 3a + p
// MakeBox (R/Instance.h:166-184): the six faces' planes and the two corners.  A face's interior test only decides
// accept / reject, and alpha = (P[p] - Q[p]) / u[p] up to a few ulps; so a hit point P that is inside [mn, mx] by more
// than 2^-30 of the coordinates' magnitude is accepted, one that is outside by as much is rejected, and only the
// sliver in between evaluates the face's own alpha / beta (render.hip box_closest has the bound).
struct BoxRec { double na[6], d[6], mn[3], mx[3]; uint32_t quad_first, pad; };


// Instance transform step (R/Instance.h:31-37 Translate, :74-112 RotateY).
enum : uint32_t { XF_TRANSLATE = 0u, XF_ROTATE_Y = 1u };
struct Xform { double a, b, c; uint32_t kind; uint32_t pad; };  // translate: offset xyz; rotate: a = sin, b = cos

// Composite leaf: [ConstantMedium] -> chain of Translate/RotateY (outermost first) -> geometry.
enum : uint32_t { GEOM_SINGLE = 0u, GEOM_SPHERES = 1u, GEOM_MSPHERES = 2u, GEOM_QUADS = 3u, GEOM_MIXED = 4u,
                  GEOM_BVH = 5u,    // `first` = root of a sub-BVH (in nodes[]) over the group's primitives
                  GEOM_BOX = 6u };  // six axis-aligned quads that form a MakeBox box: `first` = index into boxes
constexpr uint32_t kSubBvhMinPrims = 16;  // groups at least this large get a sub-BVH
struct ObjectRec {
    uint32_t geom_kind;  // GEOM_*
    uint32_t first;      // SINGLE: a prim ref; homogeneous: first prim index; MIXED: first entry of items[]
    uint32_t count;
    uint32_t xf_first, xf_count;
    uint32_t medium;     // index into media or kNone
    uint32_t coop_first; // GEOM_BVH over static spheres only: their rows are spheres[coop_first .. coop_first + count), so a wave
                         // may scan them together instead of walking the sub-BVH lane by lane; kNone otherwise
    uint32_t coop_boxes; // ... and group_boxes[coop_boxes + g] bounds rows 16 g .. 16 g + 15 of them (a cull for that scan)
};
// Bounding box of sixteen consecutive rows of a cooperative sphere group, padded outwards (a conservative cull: a ray
// that could hit one of the sixteen always passes the slab test).  The rows are in the sub-BVH's leaf order, so
// consecutive rows are neighbours in space.
struct GroupBox { double lo[3], hi[3]; };
constexpr uint32_t kCoopGroup = 16;
// R/ConstantMedium.h:39-44.  When the boundary is a lone Sphere without transforms (both media of the Book-2 final scene),
// its row is repeated here so that the medium test needs no further table: sphere = its index in spheres[], else kNone.
struct MediumRec { double neg_inv_density; uint32_t phase_mat; uint32_t sphere; double cx, cy, cz, r2; };

// General nesting.  The reference's wrappers take any Hittable* (R/Instance.h:31,74, R/ConstantMedium.h:32,39,
// R/HittableList.h:21, R/BvhNode.h:50).  Whatever ObjectRec cannot express -- a medium under a transform or inside another
// medium, lists / BVHs of composites inside an instance, a BvhNode inside a list -- stays a tree of these records and is
// evaluated by an explicit-stack interpreter (render.hip tree_hit) that makes the reference's calls in the reference's order.
// `chain` = the transforms the world ray has gone through when this node's Hit is called (outermost first, a private
// contiguous run of xforms[]): the local ray is always recomputed from the world ray, never un-transformed.
enum : uint32_t { TN_PRIM = 0u, TN_TRANSLATE = 1u, TN_ROTATE_Y = 2u, TN_MEDIUM = 3u, TN_LIST = 4u, TN_BVH = 5u };
struct TreeNodeRec {
    uint32_t kind;
    uint32_t a;  // PRIM: primitive ref; TRANSLATE / ROTATE_Y / MEDIUM: child node; LIST: first entry of tree_items[]; BVH: root in tree_bvh[]
    uint32_t b;  // MEDIUM: index into media; LIST: number of children
    uint32_t chain_first, chain_count;
    uint32_t pad0, pad1, pad2;
};
constexpr uint32_t kTreeMaxDepth = 16;      // frames of the interpreter's stack (the reference's recursion is bounded by its 32 KiB stack)
constexpr uint32_t kTreeObjBit = 0x80000000u;  // HitInfo::obj of a hit inside a tree: this bit | node whose chain applies

// Threaded BVH node, preorder.  Inner node: a = b = REF_INNER marker, first child = this + 1.
// Bottom node (span 1 or 2, R/BvhNode.h:63-72): a, b = leaf refs (a == b for span 1).
struct BvhNodeRec { double xlo, xhi, ylo, yhi, zlo, zhi; uint32_t a, b, escape, pad; };

// The library's own tree for BVH worlds of primitives only.  Such a world's closest hit does not depend on how it is
// searched (no leaf draws random numbers: the reference's BVH == list invariant, Docs 2-3 BVH :733,:772), so instead of
// the reference's median-split tree in its fixed visiting order (36 node visits and 5 leaf tests per ray on the random-
// spheres scene) the kernel may walk a surface-area-heuristic tree near child first.  Still a stackless walk over a flat
// array: every node carries, for each of the eight sign octants of the ray direction, the node to go to when its box is hit
// (the near child) and when it is not / when its subtree is done (the escape) -- the near-first order depends only on
// the octant.  88 bytes: the box, two leaf refs (bottom nodes hold one or two primitives; a = REF_INNER marker otherwise),
// 8 x {hit, escape} as 16-bit node indices (escape 0xFFFF = the walk is over; hit of a bottom node: kFastBottom | kind << 12).
struct FastNodeRec {
    double xlo, xhi, ylo, yhi, zlo, zhi;
    uint32_t a, b;
    uint16_t link[8][2];
};
constexpr uint32_t kFastEnd = 0xFFFFu;
// The row as the kernels read it (built at upload time, device_scene.cpp): the box in fp32, rounded OUTWARDS and widened by
// 2^-19 of the scene's reach.  A node's box only ever prunes -- hits are decided by the leaf tests, in fp64, with the
// reference's arithmetic -- so a box that is a little too large changes no frame; a conservative fp32 slab test
// (render.hip box_test_f) then costs half the issue slots of the fp64 one.  68 bytes: 17 words, an odd stride over the
// LDS banks.  a, b and the links as in FastNodeRec.
struct FastNodeF { float box[6]; uint32_t a, b; uint16_t link[8][2]; uint32_t pad; };
constexpr uint32_t kFastBottom = 0x8000u;    // hit link of a bottom node: this bit | kind of its first leaf << 12 (the lane parks there)
constexpr uint32_t kFastMaxNodes = 0x8000u;  // node indices stay below kFastBottom

// The same tree over the SURFACE leaves of a BVH world that also holds ConstantMedium leaves (the Book-2 final scene: 400
// boxes, spheres, an instanced cluster, and two media).  Only a medium's test draws random numbers, so only for the media
// does it matter what the reference has found before it reaches them: the closest hit among the leaves that PRECEDE the
// medium in the reference's fixed visiting order (world_items order; the BVH only prunes).  The surfaces between two
// media are one segment whose closest hit may be searched in any order.  A ray therefore walks the library's tree once
// per segment, restricted to the segment's range of leaf positions -- and a medium whose boundary the ray's line cannot
// meet (slab test of its padded box) splits nothing: for most rays of that scene, one walk (render.hip seg_advance).
// FastOrder: per node of fast_nodes, the range of leaf positions below it and the positions of a bottom node's two leaves.
struct FastOrder { uint16_t omin, omax, oa, ob; };
// One medium leaf, in visiting order: its position among the world's leaves, its object record, whether the reference calls
// it twice in a row (the duplicated leaf of a span-1 node, R/BvhNode.h:63-67), and its bounding box padded outwards.
//   candidates: the surface leaves whose boxes meet the padded box -- all there is to hit for a ray whose remaining stretch lies inside
//   the box (a ray scattered inside the medium: a third of all rays of the Book-2 final scene): seg_cand[cand_first .. + cand_count),
//   each with its position; cand_count = kNone when there are too many of them or one is an instance (such rays then walk the tree).
struct SegMedium { double lo[3], hi[3]; uint32_t order, object, twice, cand_first, cand_count, pad; float fbox[6]; };  // fbox: lo/hi as FastNodeF::box
struct SegCandidate { uint32_t ref, order; };
constexpr uint32_t kSegMaxCandidates = 12;
constexpr uint32_t kSegMaxMedia = 4;
constexpr uint32_t kSegEnd = 0xFFFFu;  // "to the end of the list" as an upper bound of a segment

// Materials (R/Material.h, R/Metal.h, R/Dielectric.h)
enum : uint32_t { MAT_LAMBERTIAN = 0u, MAT_METAL = 1u, MAT_DIELECTRIC = 2u, MAT_DIFFUSE_LIGHT = 3u, MAT_ISOTROPIC = 4u };
// metal: rgb + fuzz; dielectric: p = ior.
// needs_uv: the texture tree contains an ImageTexture, the only reader of HitRecord::U/V (R/Texture.h:110-133).
// tex_inline: the texture is a SolidColor (1) or a CheckerTexture of two SolidColors (2), resolved by the host
// into even/odd/inv_scale so that shading needs no dependent texture-table loads; 0 = walk the texture table.
struct MaterialRec {
    double r, g, b, p;
    uint32_t kind, tex, needs_uv, tex_inline;
    double even[3], odd[3], inv_scale;
    double pad;
};

// Textures (R/Texture.h)
enum : uint32_t { TEX_SOLID = 0u, TEX_CHECKER = 1u, TEX_IMAGE = 2u, TEX_NOISE = 3u };
struct TextureRec { double r, g, b, s; uint32_t kind, a, b_, pad; };
// solid: rgb; checker: s = 1/scale, a = even tex, b_ = odd tex; image: a = image index; noise: s = scale, a = perlin index
struct ImageRec { uint64_t offset; int32_t width, height; };
struct PerlinRec { double vec[256][3]; int32_t perm_x[256], perm_y[256], perm_z[256]; };  // R/Perlin.h:82-85

// Camera (R/Camera.h:92-101)
struct CameraRec {
    double bg[3], origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    double lens_radius, time0, time1;
};

enum : uint32_t { WORLD_BVH = 0u, WORLD_LIST = 1u };

// What the kernel receives (by value).  All pointers are device pointers.
struct DeviceScene {
    const SphereGeom *spheres;
    const SphereScanRow *sphere_scan;  // parallel to spheres
    double scan_reach;                 // max over spheres of |centre| + radius (bounds the filter's rounding error)
    const SphereScanPair *sphere_scan32;  // pairs (2 i, 2 i + 1) of the same list
    double scan_reach32;               // max of |centre| + radius over the spheres the fp32 filter decides (the others always pass)
    const SphereAux *sphere_aux;
    const MSphereGeom *mspheres;
    const SphereAux *msphere_aux;
    const QuadGeom *quads;
    const AAQuad *quad_aa;        // parallel to quads
    const BoxRec *boxes;
    const uint32_t *quad_mat;
    const ObjectRec *objects;
    const uint32_t *items;        // GEOM_MIXED entries (prim refs)
    const Xform *xforms;
    const MediumRec *media;
    const GroupBox *group_boxes;
    const BvhNodeRec *nodes;
    const FastNodeRec *fast_nodes;  // nullptr unless the world has a library tree (see FastNodeRec, FastOrder)
    const FastNodeF *fast_rows;     // the same nodes as the kernels stage them (FastNodeF)
    uint32_t n_fast_nodes;
    const FastOrder *fast_order;    // SCENE_SEGMENTED: parallel to fast_nodes
    const SegMedium *seg_media;     // SCENE_SEGMENTED: the world's medium leaves in visiting order
    const SegCandidate *seg_cand;   // SCENE_SEGMENTED: candidate leaves of the media (SegMedium::cand_first)
    uint32_t n_seg_cand;
    uint32_t n_seg_media;
    const TreeNodeRec *tree_nodes;
    const uint32_t *tree_items;   // children of TN_LIST nodes (tree node indices)
    const BvhNodeRec *tree_bvh;   // threaded nodes of the BvhNodes inside trees (a table of their own: nodes[] starts with the world's)
    const uint32_t *world_items;  // WORLD_LIST: leaf refs in list order
    const MaterialRec *materials;
    const TextureRec *textures;
    const ImageRec *images;
    const unsigned char *image_bytes;
    const PerlinRec *perlin;
    const CameraRec *camera;
    // BVH worlds whose leaves are all spheres / unit-time moving spheres (SCENE_WORLD_MSPHERES): the leaves once more, in
    // leaf order (entry k is world_items[k]), as seven planes of ms_padded doubles (c0x, c0y, c0z, dcx, dcy, dcz, r2; plane p
    // at ms_planes + p * ms_padded) for the cooperative scan of thin waves (render.hip scan_grouped_ms).  nullptr otherwise.
    const double *ms_planes;
    uint32_t ms_padded;
    uint32_t world_kind;
    uint32_t n_world_items;
    uint32_t n_nodes;        // all threaded nodes: the world's first, then sub-BVHs of large groups
    uint32_t n_world_nodes;
    uint32_t scan_cost;      // what testing every world leaf once costs, in half sphere tests
    uint32_t n_spheres, n_mspheres, n_quads, n_objects, n_boxes, n_xforms;
    uint32_t n_media, n_materials, n_perlin, n_group_boxes;
    // Tables a leaf test or the shading chases through -- object record -> transforms -> box / quad rows, medium rows,
    // material rows, Perlin tables -- are staged in LDS behind the node rows where they fit.  Byte offsets into the
    // dynamic LDS block, set by the launcher per table; kNone = read the global table.
    uint32_t lds_quad_aa, lds_boxes, lds_objects, lds_xforms, lds_media, lds_materials, lds_perlin, lds_spheres_tab, lds_group_boxes,
        lds_mspheres, lds_msphere_aux, lds_sphere_aux;  // the primitive tables of a sphere world (library-tree kernel, one workgroup per CU)
    uint32_t lds_fast_order, lds_seg_media, lds_seg_cand;  // segmented walk: always staged (render.hip launch_one)
    uint32_t lds_park;  // parked path state of the instanced-list kernel (render.hip Traits::PARK)
    uint32_t flags;
};

enum : uint32_t {
    SCENE_HAS_MEDIA = 1u,
    SCENE_LIST_ALL_SPHERES = 2u,  // WORLD_LIST whose leaves are spheres 0..n-1 in order (config C2 fast path)
    SCENE_RICH_TEXTURES = 4u,     // some texture is an ImageTexture or NoiseTexture
    SCENE_MS_UNIT_TIME = 8u,      // every moving-sphere row has time0 = 0, time1 - time0 = 1: frac == ray time
    SCENE_HAS_TREES = 32u,        // some leaf is a REF_TREE: rendered by the nested instantiations
    SCENE_WORLD_MSPHERES = 16u,   // WORLD_BVH of spheres / unit-time moving spheres only: ms_planes is filled
    SCENE_SEGMENTED = 64u,        // WORLD_BVH with composite leaves: fast_nodes / fast_order / seg_media describe the segmented walk
};

} // namespace rtow
