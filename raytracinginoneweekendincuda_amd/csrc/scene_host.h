// scene_host.h -- host-side object graph behind the construction API, and its flattened form.
// Internal to librtow_hip.so.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "flat_scene.h"
#include "rng.h"

namespace rtow {

struct D3 {
    double x, y, z;
};

struct Box {  // R/AABB.h:16-22: three closed intervals
    double lo[3], hi[3];
};

enum class HKind : uint8_t { Sphere, MovingSphere, Quad, Translate, RotateY, List, Bvh, Medium };

struct HostHittable {
    HKind kind;
    Box box;
    uint32_t material = 0;             // handle (1-based) for primitives; phase function for media
    D3 c0{}, c1{};                     // sphere centre(s)
    double t0 = 0, t1 = 0, radius = 0;
    D3 q{}, u{}, v{}, w{}, normal{};   // quad
    double plane_d = 0;
    uint32_t child = 0;                // instance / medium: wrapped hittable handle
    D3 offset{};
    double sin_t = 0, cos_t = 0;
    std::vector<uint32_t> items;       // list children / bvh leaves (in sorted order, after build)
    double neg_inv_density = 0;
    // bvh topology (built at construction): nodes in preorder
    struct TreeNode {
        Box box;
        int left, right;               // child tree-node indices, or -1
        uint32_t leaf_a, leaf_b;       // hittable handles when bottom node, else 0
    };
    std::vector<TreeNode> tree;
};

struct HostMaterial {
    uint32_t kind;
    uint32_t texture = 0;  // handle (1-based) or 0
    D3 albedo{};
    double p = 0;
};

struct HostTexture {
    uint32_t kind;
    D3 color{};
    double s = 0;
    uint32_t a = 0, b = 0;  // checker: even/odd handles; image: image index; noise: perlin index
};

struct FlatScene {
    std::vector<SphereGeom> spheres;
    std::vector<SphereScanRow> sphere_scan;  // parallel to spheres
    double scan_reach = 0.0;
    std::vector<SphereScanPair> sphere_scan32;  // pairs of sphere_scan rows in fp32
    double scan_reach32 = 0.0;
    std::vector<SphereAux> sphere_aux;
    std::vector<MSphereGeom> mspheres;
    std::vector<double> ms_planes;       // SCENE_WORLD_MSPHERES: seven planes of ms_padded doubles (see DeviceScene)
    uint32_t ms_padded = 0;
    std::vector<SphereAux> msphere_aux;
    std::vector<QuadGeom> quads;
    std::vector<AAQuad> quad_aa;
    std::vector<BoxRec> boxes;
    std::vector<uint32_t> quad_mat;
    std::vector<ObjectRec> objects;
    std::vector<uint32_t> items;
    std::vector<Xform> xforms;
    std::vector<MediumRec> media;
    std::vector<GroupBox> group_boxes;
    std::vector<TreeNodeRec> tree_nodes;
    std::vector<uint32_t> tree_items;
    std::vector<BvhNodeRec> tree_bvh;
    std::vector<BvhNodeRec> nodes;
    std::vector<FastNodeRec> fast_nodes;
    std::vector<FastOrder> fast_order;   // SCENE_SEGMENTED
    std::vector<SegMedium> seg_media;    // SCENE_SEGMENTED
    std::vector<SegCandidate> seg_cand;  // SCENE_SEGMENTED
    std::vector<uint32_t> world_items;   // leaf refs in final order (both world kinds)
    std::vector<Box> leaf_boxes;         // introspection
    std::vector<int> leaf_kinds;         // introspection: the leaf's kind as constructed (0 sphere, 1 moving sphere, 2 quad, 3 composite)
    std::vector<MaterialRec> materials;
    std::vector<TextureRec> textures;
    std::vector<ImageRec> images;
    std::vector<unsigned char> image_bytes;
    std::vector<PerlinRec> perlin;
    uint32_t world_kind = WORLD_BVH;
    uint32_t flags = 0;
    uint32_t n_world_nodes = 0;
    uint32_t scan_cost = 0;  // cost of testing every world leaf once (scene_builder.cpp), for the scan-or-walk choice
};

struct DeviceTables;  // device_scene.cpp

struct SceneImpl {
    std::vector<HostHittable> hittables;  // handle = index + 1
    std::vector<HostMaterial> materials;
    std::vector<HostTexture> textures;
    std::vector<ImageRec> images;
    std::vector<unsigned char> image_bytes;
    std::vector<PerlinRec> perlin;
    uint32_t world = 0;
    bool has_camera = false;
    CameraRec camera{};
    bool committed = false;
    FlatScene flat;
    std::vector<DeviceTables *> device;  // one per device ordinal, lazily
    // Every change the device copies depend on (a commit, a new camera) bumps `generation`; rt_scene_upload replaces a
    // device's tables when they are older.  `launches_in_flight` counts rt_render_launch calls not yet finished: the
    // scene may not be changed while a kernel may still be reading its tables.
    uint64_t generation = 0;
    int launches_in_flight = 0;
    std::vector<void *> films_in_flight;  // the FilmImpl of every launch counted above (device_scene.cpp)
    uint32_t options = 0;                 // RT_SCENE_* (rt_scene_set_options), read by the next rt_scene_commit

    ~SceneImpl();
};

struct RngImpl {
    Xorwow state;
};

// error plumbing (thread-local message)
void set_error(const std::string &msg);
int fail(int status, const std::string &msg);

// host jump table for curand_init's sequence skip (built once, on first use)
const uint32_t *host_jump_table();

// flattening (scene_builder.cpp)
int flatten_scene(SceneImpl &s);

// device side (device_scene.cpp)
void release_device_tables(DeviceTables *t);
// A scene that goes away while launches are in flight: wait for each of them and make their films forget the scene
// (rt_scene_destroy; the films stay valid, their rt_render_finish then only reports).
void wait_for_films_in_flight(SceneImpl &s);

} // namespace rtow
