// examples/main.cpp — the reference's src/main.rs:34-56 written against the C++ host mirror: 100x100 frame, 2 spheres,
// 1 point + 1 ambient light, render(1) -> result.png in the CWD.  Usage: ./main [width height [depth]]
// RT_DEVICES=0,1,2,3 (or 0,0,0 for virtual bands on one GPU) renders through a multi-device context (rt_ctx_create_multi).
#include <cstdio>
#include <cstdlib>

#include "raytracing.hpp"

using namespace society;

int main(int argc, char** argv)
{
    const uint32_t w = argc > 2 ? (uint32_t)std::atoi(argv[1]) : 100u;
    const uint32_t h = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 100u;
    const size_t depth = argc > 3 ? (size_t)std::atoi(argv[3]) : 1u;
    try {
        Framebuffer frame(w, h, Camera());  // main.rs:35
        std::vector<Element> elements;
        elements.push_back(Element::new_unzise(Sphere(Vec3(1.0f, -1.0f, -1.0f), 0.5f), Vec3(1.0f, 0.0f, 0.0f)));  // main.rs:39-42
        elements.push_back(Element::new_unzise(Sphere(Vec3(1.0f, 0.0f, -2.0f), 0.5f), Vec3(0.0f, 1.0f, 0.0f)));   // main.rs:43-46
        std::vector<DynLight> lights;
        lights.push_back(Point::new_unsize(Vec3::ZERO(), Vec3(1.0f, 1.0f, 1.0f), 0.5f));  // main.rs:49
        lights.push_back(Ambient::new_unsize(0.1f * Vec3(1.0f, 1.0f, 1.0f)));              // main.rs:50
        std::vector<int32_t> devices;
        if (const char* env = std::getenv("RT_DEVICES"))
            for (const char* p = env; *p;) {
                char* end = nullptr;
                const long d = std::strtol(p, &end, 10);
                if (end == p) break;
                devices.push_back((int32_t)d);
                p = *end == ',' ? end + 1 : end;
            }
        if (devices.empty()) devices.push_back(0);
        Renderer renderer(std::move(frame), std::move(elements), std::move(lights), devices);
        renderer.render(depth);  // main.rs:54
        size_t lit = 0;
        for (size_t i = 0; i + 2 < renderer.last_frame.size(); i += 3)
            lit += (renderer.last_frame[i] | renderer.last_frame[i + 1] | renderer.last_frame[i + 2]) != 0;
        std::printf("wrote result.png (%ux%u, depth %zu, %zu lit pixels)\n", w, h, depth, lit);
    } catch (const Error& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
