// rtow -- thin host executable over the C-ABI; the MI355X counterpart of the reference's main()
// (R/kernel.cu:570-742): same defaults (1440x720, sceneId 9, spp rule, seed 1984, depth 50), same
// stderr lines, same output.ppm, exit code 99 on a device error.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../include/rtow.h"

// Multi-GPU frame: rows dealt to the N GPUs of the node in 8-row stripes (rank = stripe % N), every GPU renders its
// stripes into a compact buffer, ONE ncclGather over xGMI brings them to GPU 0, the host de-interleaves.
// Single process, one HIP stream and one RCCL communicator per device.
static int render_multi_gpu(rt_scene *scene, rt_render_params base, int n_gpus, double *frame, rt_render_stats *total,
                            double *gather_seconds)
{
    const int W = base.width, H = base.height, stripe = 8;
    int rows_max = 0;
    for (int r = 0; r < n_gpus; r++) {
        int n = rt_stripe_rows(H, stripe, r, n_gpus, nullptr, 0);
        if (n > rows_max) rows_max = n;
    }
    const size_t count = (size_t)rows_max * W * 3;  // doubles per rank
    std::vector<rt_film *> films(n_gpus, nullptr);
    std::vector<double *> send(n_gpus, nullptr);
    std::vector<hipStream_t> streams(n_gpus, nullptr);
    std::vector<ncclComm_t> comms(n_gpus);
    std::vector<int> devs(n_gpus);
    double *recv = nullptr;
    auto hip_ok = [](hipError_t e, const char *what) {
        if (e != hipSuccess) {
            std::fprintf(stderr, "HIP error = %u at '%s'\n", (unsigned)e, what);
            std::exit(99);
        }
    };
    auto nccl_ok = [](ncclResult_t r, const char *what) {
        if (r != ncclSuccess) {
            std::fprintf(stderr, "RCCL error = %d (%s) at '%s'\n", (int)r, ncclGetErrorString(r), what);
            std::exit(99);
        }
    };
    for (int r = 0; r < n_gpus; r++) {
        devs[r] = r;
        hip_ok(hipSetDevice(r), "hipSetDevice");
        hip_ok(hipStreamCreateWithFlags(&streams[r], hipStreamNonBlocking), "hipStreamCreate");
        hip_ok(hipMalloc((void **)&send[r], count * sizeof(double)), "hipMalloc(send)");
        hip_ok(hipMemsetAsync(send[r], 0, count * sizeof(double), streams[r]), "hipMemsetAsync");
        films[r] = rt_film_create(r, W, H, stripe, r, n_gpus);
        if (!films[r] || rt_film_bind_pixels(films[r], send[r]) != RT_OK) return 1;
        if (rt_scene_upload(scene, r) != RT_OK) return 1;
    }
    hip_ok(hipSetDevice(0), "hipSetDevice(0)");
    hip_ok(hipMalloc((void **)&recv, count * n_gpus * sizeof(double)), "hipMalloc(recv)");
    nccl_ok(ncclCommInitAll(comms.data(), n_gpus, devs.data()), "ncclCommInitAll");

    for (int r = 0; r < n_gpus; r++) {  // all GPUs render concurrently
        rt_render_params p = base;
        p.device = r;
        p.rank = r;
        p.world_size = n_gpus;
        p.stripe_rows = stripe;
        p.stream = streams[r];
        if (rt_render_launch(scene, films[r], &p) != RT_OK) return 1;
    }
    auto t0 = std::chrono::steady_clock::now();
    nccl_ok(ncclGroupStart(), "ncclGroupStart");
    for (int r = 0; r < n_gpus; r++)
        nccl_ok(ncclGather(send[r], r == 0 ? recv : nullptr, count, ncclDouble, 0, comms[r], streams[r]), "ncclGather");
    nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
    std::memset(total, 0, sizeof *total);
    for (int r = 0; r < n_gpus; r++) {
        rt_render_stats st{};
        if (rt_render_finish(scene, films[r], &st) != RT_OK) return 1;
        hip_ok(hipSetDevice(r), "hipSetDevice");
        hip_ok(hipStreamSynchronize(streams[r]), "hipStreamSynchronize");
        total->samples += st.samples;
        total->rays += st.rays;
        if (st.seconds_render + st.seconds_seed > total->seconds_render) total->seconds_render = st.seconds_render + st.seconds_seed;
        total->kernel_vgprs = st.kernel_vgprs;
        total->lds_bytes = st.lds_bytes;
        total->kernel_kind = st.kernel_kind;
    }
    *gather_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<double> gathered(count * n_gpus);
    hip_ok(hipSetDevice(0), "hipSetDevice(0)");
    hip_ok(hipMemcpy(gathered.data(), recv, gathered.size() * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
    int rc = rt_deinterleave(gathered.data(), W, H, stripe, n_gpus, count, frame);
    for (int r = 0; r < n_gpus; r++) {
        hipSetDevice(r);
        ncclCommDestroy(comms[r]);
        rt_film_destroy(films[r]);
        hipFree(send[r]);
        hipStreamDestroy(streams[r]);
    }
    hipSetDevice(0);
    hipFree(recv);
    return rc;
}

// Binary PPM (P6, maxval 255) -> RGB bytes, row 0 = top: the layout RtwImage hands to ImageTexture (R/RtwImage.h:51-92).
static bool read_p6(const std::string &path, std::vector<unsigned char> &rgb, int &w, int &h)
{
    FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) return false;
    auto token = [&](char *buf, size_t cap) {  // header tokens, '#' comments skipped
        size_t n = 0;
        int c = std::fgetc(fp);
        while (c != EOF) {
            if (c == '#') {
                while (c != EOF && c != '\n') c = std::fgetc(fp);
            } else if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
                c = std::fgetc(fp);
            } else {
                break;
            }
        }
        while (c != EOF && !(c == ' ' || c == '\t' || c == '\n' || c == '\r') && n + 1 < cap) {
            buf[n++] = (char)c;
            c = std::fgetc(fp);
        }
        buf[n] = 0;
        return n > 0;
    };
    char magic[8], sw[16], sh[16], smax[16];
    bool ok = token(magic, sizeof magic) && std::strcmp(magic, "P6") == 0 && token(sw, sizeof sw) && token(sh, sizeof sh) &&
              token(smax, sizeof smax) && std::atoi(smax) == 255;
    if (ok) {
        w = std::atoi(sw);
        h = std::atoi(sh);
        ok = w > 0 && h > 0 && (long long)w * h < (1ll << 28);
    }
    if (ok) {
        rgb.resize((size_t)w * h * 3);
        ok = std::fread(rgb.data(), 1, rgb.size(), fp) == rgb.size();
    }
    std::fclose(fp);
    return ok;
}

static int die(const char *what)
{
    std::fprintf(stderr, "%s: %s\n", what, rt_last_error());
    return 99;  // R/kernel.cu:29-40: checkCudaErrors -> exit(99)
}

int main(int argc, char **argv)
{
    int width = 1440, height = 720, scene_id = 9, spp = -1, depth = 50, world_kind = 0, variant = 1, device = 0, gpus = 0;
    unsigned long long seed = 1984;
    unsigned flags = 0;          // RT_FLAG_* bits (schedulers and the opt-in list acceleration; none changes the picture)
    std::string out = "output.ppm";
    std::string earth_path;      // decoded 8-bit sRGB pixels of the earth texture (P6): converted like RtwImage::Load does
    bool earth_is_bytes = false; // --earth-bytes: the file already holds what RtwImage::Load hands to ImageTexture
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
        else if (const char *v = val("--gpus")) gpus = std::atoi(v);  // >= 1: stripe the frame over that many GPUs + one RCCL gather
        else if (const char *v = val("--flags")) flags = (unsigned)std::strtoul(v, nullptr, 0);
        else if (a == "--accelerate-lists") flags |= RT_FLAG_ACCELERATE_LISTS;
        else if (const char *v = val("--output")) out = v;
        else if (const char *v = val("--earth")) earth_path = v;
        else if (const char *v = val("--earth-bytes")) {
            earth_path = v;
            earth_is_bytes = true;
        } else {
            std::fprintf(stderr,
                         "usage: rtow [--scene 0..11] [--width W] [--height H] [--spp N] [--depth D] [--seed S]\n"
                         "            [--world bvh|list] [--variant strict|fast] [--device N] [--gpus N] [--output file.ppm]\n"
                         "            [--earth earthmap.jpg|decoded.ppm | --earth-bytes texture.ppm] [--accelerate-lists] [--flags N]\n"
                         "  --earth        the texture of scenes 2 and 9: a JPEG file (default: ./earthmap.jpg, like the reference) is read as\n"
                         "                 RtwImage::Load reads it -- decoded as the reference's stb_image decodes it, bit for bit; a binary PPM\n"
                         "                 (P6) is taken as pixels some other decoder produced (libjpeg's differ from stb's in ~0.6 %% of the\n"
                         "                 bytes: near-identical texture, parity unpinned) and is linearised and re-quantised the same way\n"
                         "  --earth-bytes  binary PPM (P6) that already holds the bytes RtwImage::Load hands to ImageTexture; with the\n"
                         "                 bytes the reference's own stb_image build decodes (tests/golden/earthmap_stb.npz, written out\n"
                         "                 by tests/golden/make_earth_golden.py) this is the only input that reproduces the reference's\n"
                         "                 texture bit for bit\n"
                         "  --accelerate-lists  render a list world of primitives through the library's tree (same picture, faster)\n"
                         "  --flags        RT_FLAG_* bits of include/rtow.h (none of them changes the picture)\n");
            return 2;
        }
    }
    if (spp < 0) spp = (scene_id == 9) ? 100 : ((scene_id >= 5 && scene_id <= 8) ? 200 : 10);  // R/kernel.cu:593

    std::fprintf(stderr, "Rendering a %dx%d image with %d samples per pixel in 8x8 blocks.\n", width, height, spp);
    rt_scene *scene = rt_scene_create();
    // R/kernel.cu:656-665: scenes 2 and 9 load earthmap.jpg through stb_image.  This executable carries no JPEG decoder:
    // the decoded pixels come in as a PPM (--earth / --earth-bytes; ./earthmap.ppm is picked up like the reference picks
    // up ./earthmap.jpg).  Without one the sphere shows the reference's own fallback for a missing file, cyan
    // (R/Texture.h:113-114) -- and the picture then differs from the reference's, which ships the file.
    std::vector<unsigned char> earth;
    int earth_w = 0, earth_h = 0;
    if (scene_id == 2 || scene_id == 9) {
        if (earth_path.empty()) {  // R/kernel.cu:661: RtwImage::Load("earthmap.jpg") from the working directory
            for (const char *name : {"earthmap.jpg", "earthmap.ppm"})
                if (FILE *probe = std::fopen(name, "rb")) {
                    std::fclose(probe);
                    earth_path = name;
                    break;
                }
        }
        const bool is_jpeg = earth_path.size() > 4 && (earth_path.rfind(".jpg") == earth_path.size() - 4 || earth_path.rfind(".jpeg") == earth_path.size() - 5 ||
                                                       earth_path.rfind(".JPG") == earth_path.size() - 4);
        if (earth_path.empty()) {
            std::fprintf(stderr, "ERROR: Could not load image file 'earthmap.jpg'.\n");  // R/RtwImage.h:57; the texture renders cyan (R/Texture.h:113-114)
        } else if (is_jpeg) {
            // the whole of RtwImage::Load (JPEG decode as the reference's stb_image does it, linearisation, FloatToByte)
            unsigned char *bytes = nullptr;
            if (rt_rtwimage_load(earth_path.c_str(), &bytes, &earth_w, &earth_h) != RT_OK) {
                std::fprintf(stderr, "ERROR: Could not load image file '%s' (%s).\n", earth_path.c_str(), rt_last_error());
                earth_w = earth_h = 0;
            } else {
                earth.assign(bytes, bytes + (size_t)earth_w * earth_h * 3);
                rt_image_free(bytes);
                std::fprintf(stderr, "Loaded image '%s' (%dx%d) and uploaded to device.\n", earth_path.c_str(), earth_w, earth_h);
            }
        } else if (!read_p6(earth_path, earth, earth_w, earth_h)) {
            std::fprintf(stderr, "ERROR: Could not load image file '%s'.\n", earth_path.c_str());
            earth.clear();
            earth_w = earth_h = 0;
        } else {
            if (!earth_is_bytes) rt_rtwimage_bytes(earth.data(), earth.size(), earth.data());
            std::fprintf(stderr, "Loaded image '%s' (%dx%d) and uploaded to device.\n", earth_path.c_str(), earth_w, earth_h);
        }
    }
    if (rt_scene_build_builtin(scene, scene_id, world_kind, width, height, seed, earth.empty() ? nullptr : earth.data(), earth_w,
                               earth_h) != RT_OK)
        return die("scene");

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
    p.flags = flags;
    std::vector<double> frame((size_t)width * height * 3);
    rt_render_stats st{};
    auto t0 = std::chrono::steady_clock::now();
    double gather_s = 0.0;
    if (gpus >= 1) {
        if (render_multi_gpu(scene, p, gpus, frame.data(), &st, &gather_s) != 0) return die("render (multi-GPU)");
    } else if (rt_render(scene, &p, frame.data(), &st) != RT_OK) {
        return die("render");
    }
    double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double kern = st.seconds_seed + st.seconds_render;
    if (gpus >= 1) std::fprintf(stderr, "%d GPU(s): slowest rank %.4f s of kernels; launch-to-gathered %.4f s\n", gpus, kern, gather_s);
    std::fprintf(stderr, "took %g seconds.\n", kern);
    std::fprintf(stderr, "%.1f Msamples/s, %.1f Mray/s (kernels); %.3f s wall incl. upload/download\n",
                 st.samples / kern * 1e-6, st.rays / kern * 1e-6, wall);
    if (rt_write_ppm(out.c_str(), frame.data(), width, height) != RT_OK) return die("ppm");
    std::fprintf(stderr, "\nDone. Saved to %s\n", out.c_str());
    rt_scene_destroy(scene);
    return 0;
}
