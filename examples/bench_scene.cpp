// examples/bench_scene.cpp — native (no Python) harness over the C ABI: generates a BASELINE.json config with SCENE-GEN v1,
// renders it into a page-locked frame store and prints kernel time, end-to-end time and an FNV-1a hash of the frame (the GPU test
// compares the frame with the one of the Python path).  Usage: ./bench_scene [C3|C4|C5] [frames] [shadows 0|1] [depth] [frame.rgba]
//   ./bench_scene --scene C3   prints the generated scene as f32 bit patterns and touches no device (CPU test of SCENE-GEN in C++)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_mi355x.h"
#include "scenegen.hpp"

using namespace society;

static int fail(const char* what, rt_ctx* ctx)
{
    std::fprintf(stderr, "error: %s: %s\n", what, rt_last_error(ctx));
    return 1;
}

int main(int argc, char** argv)
{
    if (argc > 2 && !std::strcmp(argv[1], "--scene")) {
        SyntheticScene g;
        if (!config_scene(argv[2], g)) return 2;
        auto bits = [](float v) { uint32_t b; std::memcpy(&b, &v, 4); return b; };
        for (size_t k = 0; k < g.cx.size(); ++k)
            std::printf("S %08x %08x %08x %08x %08x %08x %08x\n", bits(g.cx[k]), bits(g.cy[k]), bits(g.cz[k]), bits(g.radius[k]), bits(g.rr[k]), bits(g.rg[k]), bits(g.rb[k]));
        for (size_t k = 0; k < g.px.size(); ++k)
            std::printf("L %08x %08x %08x %08x\n", bits(g.px[k]), bits(g.py[k]), bits(g.pz[k]), bits(g.intensity[k]));
        return 0;
    }
    const char* name = argc > 1 ? argv[1] : "C3";
    const int frames = argc > 2 ? std::atoi(argv[2]) : 20;
    // RT_FLAGS_EXTRA: further library flags, e.g. RT_FLAGS_EXTRA=8 = RT_CULL (same frame through the cone hierarchy: the hash must not change)
    const uint32_t extra = std::getenv("RT_FLAGS_EXTRA") ? (uint32_t)std::strtoul(std::getenv("RT_FLAGS_EXTRA"), nullptr, 0) : 0u;
    const uint32_t flags = ((argc > 3 && std::atoi(argv[3])) ? RT_SHADOWS : 0u) | extra;
    const uint32_t depth = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 1u;
    SyntheticScene sc;
    if (!config_scene(name, sc) || frames < 1) {
        std::fprintf(stderr, "usage: bench_scene [C3|C4|C5] [frames >= 1] [shadows 0|1] [depth]\n");
        return 2;
    }
    rt_ctx* ctx = nullptr;
    // RT_DEVICES=0,1,2,3,4,5,6,7 (or 0,0,0,0: virtual bands on one GPU): one context over a device list, the row-band split and the RCCL
    // gather happen inside rt_render (rt_ctx_create_multi)
    std::vector<int32_t> devices;
    if (const char* env = std::getenv("RT_DEVICES"))
        for (const char* p = env; *p;) {
            char* end = nullptr;
            const long d = std::strtol(p, &end, 10);
            if (end == p) break;
            devices.push_back((int32_t)d);
            p = *end == ',' ? end + 1 : end;
        }
    if (devices.size() > 1) {
        if (rt_ctx_create_multi(&ctx, devices.data(), (int32_t)devices.size()) != RT_OK) return fail("rt_ctx_create_multi", nullptr);
        std::printf("multi-device context: %zu row bands, root device %d\n", devices.size(), devices[0]);
    } else if (rt_ctx_create(&ctx, devices.empty() ? 0 : devices[0]) != RT_OK) return fail("rt_ctx_create", nullptr);
    const rt_objects o = sc.objects();
    const rt_lights l = sc.lights();
    if (rt_scene_upload(ctx, &o, &l) != RT_OK) return fail("rt_scene_upload", ctx);
    float fov, zn, zf, m[16];
    rt_camera_default(&fov, &zn, &zf);
    rt_camera_transform(fov, zn, zf, (float)sc.width / (float)sc.height, m);
    const size_t bytes = (size_t)sc.width * sc.height * 4;
    void* frame = nullptr;
    if (rt_host_alloc(ctx, bytes, &frame) != RT_OK) return fail("rt_host_alloc", ctx);
    if (rt_render(ctx, m, sc.width, sc.height, depth, flags, static_cast<uint8_t*>(frame)) != RT_OK) return fail("rt_render", ctx);  // warm-up
    double kernel_ms = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; ++f) {
        if (rt_render(ctx, m, sc.width, sc.height, depth, flags, static_cast<uint8_t*>(frame)) != RT_OK) return fail("rt_render", ctx);
        float ms = 0.0f;
        if (rt_last_kernel_ms(ctx, &ms) != RT_OK) return fail("rt_last_kernel_ms", ctx);
        kernel_ms += ms;
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
    uint64_t hash = 1469598103934665603ull;  // FNV-1a over the RGBA8 frame
    const uint8_t* p = static_cast<const uint8_t*>(frame);
    size_t lit = 0;
    for (size_t i = 0; i < bytes; ++i) {
        hash = (hash ^ p[i]) * 1099511628211ull;
        if ((i & 3) == 3) lit += (p[i - 3] | p[i - 2] | p[i - 1]) != 0;
    }
    std::printf("%s %ux%u spheres=%d lights=%d depth=%u shadows=%u: kernels %.4f ms/frame, render + copy to host %.4f ms/frame, "
                "%.4e primary rays/s end to end, lit pixels %zu, fnv1a %016llx\n",
                name, sc.width, sc.height, o.n, l.n, depth, flags & RT_SHADOWS, kernel_ms / frames, wall_ms,
                (double)sc.width * sc.height / (wall_ms * 1e-3), lit, (unsigned long long)hash);
    if (argc > 5) {
        FILE* f = std::fopen(argv[5], "wb");
        if (!f || std::fwrite(frame, 1, bytes, f) != bytes) { std::fprintf(stderr, "error: cannot write %s\n", argv[5]); return 1; }
        std::fclose(f);
    }
    rt_host_free(ctx, frame);
    rt_ctx_destroy(ctx);
    return 0;
}
