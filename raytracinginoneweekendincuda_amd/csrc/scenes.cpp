// scenes.cpp -- the reference's ten scenes (sceneId 0..9, R/kernel.cu:199-517) plus the two benchmark
// variants, written against the construction API exactly as a user of the reference would port
// CreateWorld.  Random draws the reference writes inside constructor argument lists are taken here
// one statement at a time, left to right.
#include <cmath>

#include "../../include/rtow.hpp"
#include "scene_host.h"

using namespace rtow_api;

namespace {

struct View {
    Vector3 from{13.0, 2.0, 3.0}, at{0.0, 0.0, 0.0};
    double vfov = 20.0, aperture = 0.0, focus = 10.0, shutter0 = 0.0, shutter1 = 0.0;
    Color background{0.70, 0.80, 1.00};
};

// sceneId 0 (R/kernel.cu:199-258); static_only = config C2 (every MovingSphere becomes a Sphere at `center`,
// the center2 draw is still consumed so both variants share one layout).
void random_spheres(Scene &w, Rng &rnd, std::vector<Hittable> &list, View &view, bool static_only)
{
    Texture checker = w.CheckerTexture(0.32, w.SolidColor(Color(0.2, 0.3, 0.1)), w.SolidColor(Color(0.9, 0.9, 0.9)));
    list.push_back(w.Sphere(Vector3(0.0, -1000.0, -1.0), 1000.0, w.Lambertian(checker)));
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            double chooseMat = rnd();
            double cx = a + 0.9 * rnd();
            double cz = b + 0.9 * rnd();
            Vector3 center(cx, 0.2, cz);
            double ddx = center.x - 4.0, ddy = center.y - 0.2, ddz = center.z - 0.0;
            if (std::sqrt(ddx * ddx + ddy * ddy + ddz * ddz) <= 0.9) continue;
            if (chooseMat < 0.8) {
                double lift = 0.5 * rnd();
                Vector3 center2(center.x + 0.0, center.y + lift, center.z + 0.0);
                float r0 = rnd(), r1 = rnd(), g0 = rnd(), g1 = rnd(), b0 = rnd(), b1 = rnd();
                Material m = w.Lambertian(Color(r0 * r1, g0 * g1, b0 * b1));  // float products, R/kernel.cu:229
                list.push_back(static_only ? w.Sphere(center, 0.2, m) : w.MovingSphere(center, center2, 0.0, 1.0, 0.2, m));
            } else if (chooseMat < 0.95) {
                double r = 0.5 * (1.0 + rnd());
                double g = 0.5 * (1.0 + rnd());
                double bl = 0.5 * (1.0 + rnd());
                double fuzz = 0.5 * rnd();
                list.push_back(w.Sphere(center, 0.2, w.Metal(Color(r, g, bl), fuzz)));
            } else {
                list.push_back(w.Sphere(center, 0.2, w.Dielectric(1.5)));
            }
        }
    }
    list.push_back(w.Sphere(Vector3(0.0, 1.0, 0.0), 1.0, w.Dielectric(1.5)));
    list.push_back(w.Sphere(Vector3(-4.0, 1.0, 0.0), 1.0, w.Lambertian(Color(0.4, 0.2, 0.1))));
    list.push_back(w.Sphere(Vector3(4.0, 1.0, 0.0), 1.0, w.Metal(Color(0.7, 0.6, 0.5), 0.0)));
    view.from = Vector3(13.0, 2.0, 3.0);
    view.vfov = 30.0;
    view.aperture = 0.1;
    view.shutter0 = 0.0;
    view.shutter1 = 1.0;
}

struct CornellMaterials {
    Material red, white, green, light;
};
CornellMaterials cornell_materials(Scene &w, double emit)
{
    CornellMaterials m;
    m.red = w.Lambertian(Color(0.65, 0.05, 0.05));
    m.white = w.Lambertian(Color(0.73, 0.73, 0.73));
    m.green = w.Lambertian(Color(0.12, 0.45, 0.15));
    m.light = w.DiffuseLight(Color(emit, emit, emit));
    return m;
}
void cornell_view(View &view)
{
    view.background = Color(0.0, 0.0, 0.0);
    view.from = Vector3(278.0, 278.0, -800.0);
    view.at = Vector3(278.0, 278.0, 0.0);
    view.vfov = 40.0;
    view.aperture = 0.0;
}
// the two instanced boxes of scenes 7 and 8 (R/kernel.cu:379-387,416-424)
void cornell_boxes(Scene &w, Material white, Hittable &box1, Hittable &box2)
{
    box1 = w.MakeBox(Point3(0, 0, 0), Point3(165, 330, 165), white);
    box1 = w.RotateY(box1, 15.0);
    box1 = w.Translate(box1, Vector3(265, 0, 295));
    box2 = w.MakeBox(Point3(0, 0, 0), Point3(165, 165, 165), white);
    box2 = w.RotateY(box2, -18.0);
    box2 = w.Translate(box2, Vector3(130, 0, 65));
}

} // namespace

extern "C" int rt_scene_build_builtin(rt_scene *s, int scene_id, int world_kind, int image_width, int image_height,
                                      uint64_t seed, const unsigned char *earth_rgb, int earth_w, int earth_h)
{
    if (!s || image_width <= 0 || image_height <= 0) return rtow::fail(RT_ERR_INVALID, "rt_scene_build_builtin: bad arguments");
    if (world_kind != 0 && world_kind != 1) return rtow::fail(RT_ERR_INVALID, "rt_scene_build_builtin: world_kind must be 0 (bvh) or 1 (list)");
    try {
        Scene w(s);
        Rng rnd(seed, 0);  // RandInit: curand_init(1984, 0, 0), R/kernel.cu:105
        std::vector<Hittable> list;
        View view;

        switch (scene_id) {
        case 0:
        case 11:
            random_spheres(w, rnd, list, view, scene_id == 11);
            break;
        case 1: {  // two checkered spheres, R/kernel.cu:260-273
            Texture checker = w.CheckerTexture(0.32, w.SolidColor(Color(0.2, 0.3, 0.1)), w.SolidColor(Color(0.9, 0.9, 0.9)));
            list.push_back(w.Sphere(Vector3(0.0, -10.0, 0.0), 10.0, w.Lambertian(checker)));
            list.push_back(w.Sphere(Vector3(0.0, 10.0, 0.0), 10.0, w.Lambertian(checker)));
            break;
        }
        case 2: {  // earth, R/kernel.cu:274-282
            Texture earth = w.ImageTexture(earth_rgb, earth_w, earth_h);
            list.push_back(w.Sphere(Vector3(0.0, 0.0, 0.0), 2.0, w.Lambertian(earth)));
            view.from = Vector3(0.0, 0.0, 12.0);
            break;
        }
        case 3: {  // perlin spheres, R/kernel.cu:283-292
            Texture pertext = w.NoiseTexture(4.0, rnd);
            list.push_back(w.Sphere(Vector3(0.0, -1000.0, 0.0), 1000.0, w.Lambertian(pertext)));
            list.push_back(w.Sphere(Vector3(0.0, 2.0, 0.0), 2.0, w.Lambertian(pertext)));
            break;
        }
        case 4: {  // quads, R/kernel.cu:293-309
            list.push_back(w.Quad(Vector3(-3, -2, 5), Vector3(0, 0, -4), Vector3(0, 4, 0), w.Lambertian(Color(1.0, 0.2, 0.2))));
            list.push_back(w.Quad(Vector3(-2, -2, 0), Vector3(4, 0, 0), Vector3(0, 4, 0), w.Lambertian(Color(0.2, 1.0, 0.2))));
            list.push_back(w.Quad(Vector3(3, -2, 1), Vector3(0, 0, 4), Vector3(0, 4, 0), w.Lambertian(Color(0.2, 0.2, 1.0))));
            list.push_back(w.Quad(Vector3(-2, 3, 1), Vector3(4, 0, 0), Vector3(0, 0, 4), w.Lambertian(Color(1.0, 0.5, 0.0))));
            list.push_back(w.Quad(Vector3(-2, -3, 5), Vector3(4, 0, 0), Vector3(0, 0, -4), w.Lambertian(Color(0.2, 0.8, 0.8))));
            view.from = Vector3(0.0, 0.0, 9.0);
            view.vfov = 80.0;
            break;
        }
        case 5: {  // simple light, R/kernel.cu:310-326
            Texture pertext = w.NoiseTexture(4.0, rnd);
            list.push_back(w.Sphere(Vector3(0.0, -1000.0, 0.0), 1000.0, w.Lambertian(pertext)));
            list.push_back(w.Sphere(Vector3(0.0, 2.0, 0.0), 2.0, w.Lambertian(pertext)));
            Material diffLight = w.DiffuseLight(Color(4.0, 4.0, 4.0));
            list.push_back(w.Sphere(Vector3(0.0, 7.0, 0.0), 2.0, diffLight));
            list.push_back(w.Quad(Vector3(3.0, 1.0, -2.0), Vector3(2.0, 0.0, 0.0), Vector3(0.0, 2.0, 0.0), diffLight));
            view.background = Color(0.0, 0.0, 0.0);
            view.from = Vector3(26.0, 3.0, 6.0);
            view.at = Vector3(0.0, 2.0, 0.0);
            break;
        }
        case 6:
        case 7: {  // Cornell box, empty (6) / with two instanced boxes (7), R/kernel.cu:327-398
            CornellMaterials m = cornell_materials(w, 15.0);
            list.push_back(w.Quad(Vector3(555, 0, 0), Vector3(0, 555, 0), Vector3(0, 0, 555), m.green));
            list.push_back(w.Quad(Vector3(0, 0, 0), Vector3(0, 555, 0), Vector3(0, 0, 555), m.red));
            list.push_back(w.Quad(Vector3(343, 554, 332), Vector3(-130, 0, 0), Vector3(0, 0, -105), m.light));
            list.push_back(w.Quad(Vector3(0, 0, 0), Vector3(555, 0, 0), Vector3(0, 0, 555), m.white));
            list.push_back(w.Quad(Vector3(555, 555, 555), Vector3(-555, 0, 0), Vector3(0, 0, -555), m.white));
            list.push_back(w.Quad(Vector3(0, 0, 555), Vector3(555, 0, 0), Vector3(0, 555, 0), m.white));
            if (scene_id == 7) {
                Hittable box1, box2;
                cornell_boxes(w, m.white, box1, box2);
                list.push_back(box1);
                list.push_back(box2);
            }
            cornell_view(view);
            break;
        }
        case 8: {  // Cornell smoke, R/kernel.cu:399-434
            CornellMaterials m = cornell_materials(w, 7.0);
            list.push_back(w.Quad(Vector3(555, 0, 0), Vector3(0, 555, 0), Vector3(0, 0, 555), m.green));
            list.push_back(w.Quad(Vector3(0, 0, 0), Vector3(0, 555, 0), Vector3(0, 0, 555), m.red));
            list.push_back(w.Quad(Vector3(113, 554, 127), Vector3(330, 0, 0), Vector3(0, 0, 305), m.light));
            list.push_back(w.Quad(Vector3(0, 555, 0), Vector3(555, 0, 0), Vector3(0, 0, 555), m.white));
            list.push_back(w.Quad(Vector3(0, 0, 0), Vector3(555, 0, 0), Vector3(0, 0, 555), m.white));
            list.push_back(w.Quad(Vector3(0, 0, 555), Vector3(555, 0, 0), Vector3(0, 555, 0), m.white));
            Hittable box1, box2;
            cornell_boxes(w, m.white, box1, box2);
            list.push_back(w.ConstantMedium(box1, 0.01, Color(0.0, 0.0, 0.0)));
            list.push_back(w.ConstantMedium(box2, 0.01, Color(1.0, 1.0, 1.0)));
            cornell_view(view);
            break;
        }
        case 9: {  // The Next Week final scene, R/kernel.cu:435-517
            Material ground = w.Lambertian(Color(0.48, 0.83, 0.53));
            const int boxesPerSide = 20;
            for (int bi = 0; bi < boxesPerSide; bi++) {
                for (int bj = 0; bj < boxesPerSide; bj++) {
                    double wd = 100.0;
                    double x0 = -1000.0 + bi * wd;
                    double z0 = -1000.0 + bj * wd;
                    double x1 = x0 + wd;
                    double y1 = 1.0 + 100.0 * rnd();
                    double z1 = z0 + wd;
                    list.push_back(w.MakeBox(Point3(x0, 0.0, z0), Point3(x1, y1, z1), ground));
                }
            }
            list.push_back(w.Quad(Vector3(123, 554, 147), Vector3(300, 0, 0), Vector3(0, 0, 265), w.DiffuseLight(Color(7.0, 7.0, 7.0))));
            list.push_back(w.MovingSphere(Point3(400, 400, 200), Point3(430, 400, 200), 0.0, 1.0, 50.0, w.Lambertian(Color(0.7, 0.3, 0.1))));
            list.push_back(w.Sphere(Point3(260, 150, 45), 50.0, w.Dielectric(1.5)));
            list.push_back(w.Sphere(Point3(0, 150, 145), 50.0, w.Metal(Color(0.8, 0.8, 0.9), 1.0)));
            list.push_back(w.Sphere(Point3(360, 150, 145), 70.0, w.Dielectric(1.5)));
            Hittable blueBoundary = w.Sphere(Point3(360, 150, 145), 70.0, w.Dielectric(1.5));
            list.push_back(w.ConstantMedium(blueBoundary, 0.2, Color(0.2, 0.4, 0.9)));
            Hittable mistBoundary = w.Sphere(Point3(0, 0, 0), 5000.0, w.Dielectric(1.5));
            list.push_back(w.ConstantMedium(mistBoundary, 0.0001, Color(1.0, 1.0, 1.0)));
            list.push_back(w.Sphere(Point3(400, 200, 400), 100.0, w.Lambertian(w.ImageTexture(earth_rgb, earth_w, earth_h))));
            list.push_back(w.Sphere(Point3(220, 280, 300), 80.0, w.Lambertian(w.NoiseTexture(0.2, rnd))));
            Material white = w.Lambertian(Color(0.73, 0.73, 0.73));
            std::vector<Hittable> boxes2;
            for (int k = 0; k < 1000; k++) {
                double px = 165.0 * rnd();
                double py = 165.0 * rnd();
                double pz = 165.0 * rnd();
                boxes2.push_back(w.Sphere(Point3(px, py, pz), 10.0, white));
            }
            Hittable cluster = w.HittableList(boxes2);
            cluster = w.RotateY(cluster, 15.0);
            cluster = w.Translate(cluster, Vector3(-100, 270, 395));
            list.push_back(cluster);
            view.background = Color(0.0, 0.0, 0.0);
            view.from = Vector3(478.0, 278.0, -600.0);
            view.at = Vector3(278.0, 278.0, 0.0);
            view.vfov = 40.0;
            view.shutter0 = 0.0;
            view.shutter1 = 1.0;
            break;
        }
        case 10: {  // config C1 "three spheres" (Book 1 materials demo; not in the reference's CreateWorld)
            list.push_back(w.Sphere(Vector3(0.0, -100.5, -1.0), 100.0, w.Lambertian(Color(0.8, 0.8, 0.0))));
            list.push_back(w.Sphere(Vector3(0.0, 0.0, -1.2), 0.5, w.Lambertian(Color(0.1, 0.2, 0.5))));
            list.push_back(w.Sphere(Vector3(-1.0, 0.0, -1.0), 0.5, w.Dielectric(1.5)));
            list.push_back(w.Sphere(Vector3(1.0, 0.0, -1.0), 0.5, w.Metal(Color(0.8, 0.6, 0.2), 1.0)));
            view.from = Vector3(0.0, 0.0, 0.0);
            view.at = Vector3(0.0, 0.0, -1.0);
            view.vfov = 90.0;
            break;
        }
        default:
            return rtow::fail(RT_ERR_INVALID, "rt_scene_build_builtin: scene_id must be 0..11");
        }

        // R/kernel.cu:523-528: BvhNode(list, 0, i, ...) as world; "no BVH" = HittableList(list, i)
        Hittable world = world_kind == 0 ? w.BvhNode(list) : w.HittableList(list);
        w.SetWorld(world);
        w.Camera(view.from, view.at, Vector3(0.0, 1.0, 0.0), view.vfov, double(image_width) / double(image_height),
                 view.aperture, view.focus, view.shutter0, view.shutter1, view.background);
        w.Commit();
    } catch (const std::exception &e) {
        return rtow::fail(RT_ERR_INVALID, std::string("rt_scene_build_builtin: ") + e.what());
    }
    return RT_OK;
}
