// rtow.hpp -- header-only C++17 veneer over the C-ABI (rtow.h).
//
// Keeps scene code shaped like the reference's CreateWorld (R/kernel.cu:199-517): the same
// constructor names and parameter lists (Sphere(center, radius, material), Lambertian(color),
// CheckerTexture(scale, even, odd), MakeBox(a, b, material), Translate(object, offset), ...),
// returning small value handles instead of device pointers.  Ownership is the scene's: nothing
// to delete, sharing a child between two parents is fine (the reference double-frees there,
// Docs 2-10 :202-213).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "rtow.h"

namespace rtow_api {

struct Vector3 {
    double x = 0, y = 0, z = 0;
    Vector3() = default;
    Vector3(double a, double b, double c) : x(a), y(b), z(c) {}
};
using Point3 = Vector3;
using Color = Vector3;

struct Texture { rt_handle h = 0; };
struct Material { rt_handle h = 0; };
struct Hittable { rt_handle h = 0; };

class Error : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

// curandState stand-in for scene generation: Rng rng(1984, 0); float u = rng();   (RND macro, R/kernel.cu:157)
class Rng {
public:
    Rng(unsigned long long seed, unsigned long long sequence) : r_(rt_rng_create(seed, sequence)) {}
    ~Rng() { rt_rng_destroy(r_); }
    Rng(const Rng &) = delete;
    Rng &operator=(const Rng &) = delete;
    float operator()() { return rt_rng_uniform(r_); }
    rt_rng *raw() { return r_; }

private:
    rt_rng *r_;
};

class Scene {
public:
    Scene() : s_(rt_scene_create()) {}
    ~Scene() { if (own_) rt_scene_destroy(s_); }
    explicit Scene(rt_scene *borrowed) : s_(borrowed), own_(false) {}
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;
    rt_scene *raw() { return s_; }

    // textures (R/Texture.h)
    Texture SolidColor(const Color &c) { return {ok(rt_solid_color(s_, c.x, c.y, c.z))}; }
    Texture SolidColor(double r, double g, double b) { return {ok(rt_solid_color(s_, r, g, b))}; }
    Texture CheckerTexture(double scale, Texture even, Texture odd) { return {ok(rt_checker_texture(s_, scale, even.h, odd.h))}; }
    Texture ImageTexture(const unsigned char *rgb, int w, int h) { return {ok(rt_image_texture(s_, rgb, w, h))}; }
    Texture NoiseTexture(double scale, Rng &rng) { return {ok(rt_noise_texture(s_, scale, rng.raw()))}; }

    // materials (R/Material.h, R/Metal.h, R/Dielectric.h)
    Material Lambertian(const Color &c) { return {ok(rt_lambertian(s_, c.x, c.y, c.z))}; }
    Material Lambertian(Texture t) { return {ok(rt_lambertian_tex(s_, t.h))}; }
    Material Metal(const Color &c, double fuzz) { return {ok(rt_metal(s_, c.x, c.y, c.z, fuzz))}; }
    Material Dielectric(double ior) { return {ok(rt_dielectric(s_, ior))}; }
    Material DiffuseLight(const Color &c) { return {ok(rt_diffuse_light(s_, c.x, c.y, c.z))}; }
    Material DiffuseLight(Texture t) { return {ok(rt_diffuse_light_tex(s_, t.h))}; }
    Material Isotropic(const Color &c) { return {ok(rt_isotropic(s_, c.x, c.y, c.z))}; }
    Material Isotropic(Texture t) { return {ok(rt_isotropic_tex(s_, t.h))}; }

    // hittables
    Hittable Sphere(const Point3 &c, double r, Material m) { return {ok(rt_sphere(s_, c.x, c.y, c.z, r, m.h))}; }
    Hittable MovingSphere(const Point3 &c0, const Point3 &c1, double t0, double t1, double r, Material m)
    {
        return {ok(rt_moving_sphere(s_, c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, t0, t1, r, m.h))};
    }
    Hittable Quad(const Point3 &q, const Vector3 &u, const Vector3 &v, Material m)
    {
        const double qq[3] = {q.x, q.y, q.z}, uu[3] = {u.x, u.y, u.z}, vv[3] = {v.x, v.y, v.z};
        return {ok(rt_quad(s_, qq, uu, vv, m.h))};
    }
    Hittable Translate(Hittable o, const Vector3 &off) { return {ok(rt_translate(s_, o.h, off.x, off.y, off.z))}; }
    Hittable RotateY(Hittable o, double degrees) { return {ok(rt_rotate_y(s_, o.h, degrees))}; }
    Hittable MakeBox(const Point3 &a, const Point3 &b, Material m)
    {
        const double aa[3] = {a.x, a.y, a.z}, bb[3] = {b.x, b.y, b.z};
        return {ok(rt_make_box(s_, aa, bb, m.h))};
    }
    Hittable HittableList(const std::vector<Hittable> &items)
    {
        std::vector<rt_handle> hs;
        for (const auto &i : items) hs.push_back(i.h);
        return {ok(rt_hittable_list(s_, hs.data(), (int)hs.size()))};
    }
    Hittable ConstantMedium(Hittable boundary, double density, const Color &c)
    {
        return {ok(rt_constant_medium(s_, boundary.h, density, c.x, c.y, c.z))};
    }
    Hittable ConstantMedium(Hittable boundary, double density, Texture t)
    {
        return {ok(rt_constant_medium_tex(s_, boundary.h, density, t.h))};
    }
    // BvhNode(list, 0, n, ...): sorts `items` in place like the reference does with list[]
    Hittable BvhNode(std::vector<Hittable> &items)
    {
        std::vector<rt_handle> hs;
        for (const auto &i : items) hs.push_back(i.h);
        rt_handle root = ok(rt_bvh_node(s_, hs.data(), (int)hs.size()));
        for (size_t k = 0; k < hs.size(); k++) items[k].h = hs[k];
        return {root};
    }

    void SetWorld(Hittable world) { check(rt_scene_set_world(s_, world.h)); }
    void Camera(const Point3 &lookfrom, const Point3 &lookat, const Vector3 &vup, double vfov, double aspect,
                double aperture, double focusDist, double time0 = 0.0, double time1 = 0.0,
                const Color &background = Color(0.70, 0.80, 1.00))
    {
        const double f[3] = {lookfrom.x, lookfrom.y, lookfrom.z}, a[3] = {lookat.x, lookat.y, lookat.z};
        const double u[3] = {vup.x, vup.y, vup.z}, bg[3] = {background.x, background.y, background.z};
        check(rt_scene_set_camera(s_, f, a, u, vfov, aspect, aperture, focusDist, time0, time1, bg));
    }
    void Commit() { check(rt_scene_commit(s_)); }

private:
    rt_handle ok(rt_handle h)
    {
        if (!h) throw Error(rt_last_error());
        return h;
    }
    void check(int status)
    {
        if (status != RT_OK) throw Error(rt_last_error());
    }
    rt_scene *s_;
    bool own_ = true;
};

} // namespace rtow_api
