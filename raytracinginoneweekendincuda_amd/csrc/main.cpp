// rtow -- thin host executable over the C-ABI; the MI355X counterpart of the reference's main()
// (R/kernel.cu:570-742): same defaults (1440x720, sceneId 9, spp rule, seed 1984, depth 50), same
// stderr lines, same output.ppm, exit code 99 on a device error.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtow.h"

static int die(const char *what)
{
    std::fprintf(stderr, "%s: %s\n", what, rt_last_error());
    return 99;  // R/kernel.cu:29-40: checkCudaErrors -> exit(99)
}

int main(int argc, char **argv)
{
    int width = 1440, height = 720, scene_id = 9, spp = -1, depth = 50, world_kind = 0, variant = 1, device = 0;
    unsigned long long seed = 1984;
    std::string out = "output.ppm";
    for (int k = 1; k < argc; k++) {
        std::string a = argv[k];
        auto val = [&](const char *name) -> const char * {
            if (a == name && k + 1 < argc) return argv[++k];
            return nullptr;
        };
        if (const char *v = val("--width")) width = std::atoi(v);
        else if (const char *v = val("--height")) height = std::atoi(v);
        else if (const char *v = val("--scene")) scene_id = std::atoi(v);
        else if (const char *v = val("--spp")) spp = std::atoi(v);
        else if (const char *v = val("--depth")) depth = std::atoi(v);
        else if (const char *v = val("--seed")) seed = std::strtoull(v, nullptr, 10);
        else if (const char *v = val("--world")) world_kind = std::strcmp(v, "list") == 0 ? 1 : 0;
        else if (const char *v = val("--variant")) variant = std::strcmp(v, "strict") == 0 ? 0 : 1;
        else if (const char *v = val("--device")) device = std::atoi(v);
        else if (const char *v = val("--output")) out = v;
        else {
            std::fprintf(stderr,
                         "usage: rtow [--scene 0..11] [--width W] [--height H] [--spp N] [--depth D] [--seed S]\n"
                         "            [--world bvh|list] [--variant strict|fast] [--device N] [--output file.ppm]\n");
            return 2;
        }
    }
    if (spp < 0) spp = (scene_id == 9) ? 100 : ((scene_id >= 5 && scene_id <= 8) ? 200 : 10);  // R/kernel.cu:593

    std::fprintf(stderr, "Rendering a %dx%d image with %d samples per pixel in 8x8 blocks.\n", width, height, spp);
    rt_scene *scene = rt_scene_create();
    if (rt_scene_build_builtin(scene, scene_id, world_kind, width, height, seed, nullptr, 0, 0) != RT_OK) return die("scene");

    rt_render_params p{};
    p.width = width;
    p.height = height;
    p.samples_per_pixel = spp;
    p.max_depth = depth;
    p.seed = seed;
    p.stripe_rows = 8;
    p.rank = 0;
    p.world_size = 1;
    p.variant = variant;
    p.device = device;
    std::vector<double> frame((size_t)width * height * 3);
    rt_render_stats st{};
    auto t0 = std::chrono::steady_clock::now();
    if (rt_render(scene, &p, frame.data(), &st) != RT_OK) return die("render");
    double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double kern = st.seconds_seed + st.seconds_render;
    std::fprintf(stderr, "took %g seconds.\n", kern);
    std::fprintf(stderr, "%.1f Msamples/s, %.1f Mray/s (kernels); %.3f s wall incl. upload/download\n",
                 st.samples / kern * 1e-6, st.rays / kern * 1e-6, wall);
    if (rt_write_ppm(out.c_str(), frame.data(), width, height) != RT_OK) return die("ppm");
    std::fprintf(stderr, "\nDone. Saved to %s\n", out.c_str());
    rt_scene_destroy(scene);
    return 0;
}
