// scenegen.hpp — SCENE-GEN v1 in C++ (the synthetic scenes of configs C3 / C4 / C5), bit-identical to
// raytracing_amd/scenegen.py, which is the definition: PCG32 XSH-RR (multiplier 6364136223846793005, stream 54), seed 42,
// u() = (next_u32() >> 8) * 2^-24 as f32, every derived value one f32 operation per step in the order written there.
// Build with -ffp-contract=off so that no product/sum pair is fused.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rt_mi355x.h"

namespace society {

class Pcg32 {
public:
    explicit Pcg32(uint64_t seed = 42, uint64_t stream = 54) : state_(0), inc_((stream << 1) | 1u)
    {
        next_u32();
        state_ += seed;
        next_u32();
    }
    uint32_t next_u32()
    {
        const uint64_t old = state_;
        state_ = old * 6364136223846793005ull + inc_;
        const uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
        const uint32_t rot = (uint32_t)(old >> 59);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    float u() { return (float)(next_u32() >> 8) * 5.9604644775390625e-08f; }  // 2^-24

private:
    uint64_t state_, inc_;
};

// SoA storage + the rt_objects / rt_lights views over it (valid while the SyntheticScene lives)
struct SyntheticScene {
    uint32_t width = 0, height = 0;
    std::vector<float> cx, cy, cz, radius, rr, rg, rb, px, py, pz, lr, lg, lb, intensity;
    std::vector<uint8_t> lkind;
    rt_objects objects() const { return rt_objects{(int32_t)cx.size(), nullptr, cx.data(), cy.data(), cz.data(), radius.data(), rr.data(), rg.data(), rb.data(), nullptr, nullptr, nullptr}; }
    rt_lights lights() const { return rt_lights{(int32_t)px.size(), lkind.data(), px.data(), py.data(), pz.data(), lr.data(), lg.data(), lb.data(), intensity.data()}; }
};

inline SyntheticScene generate_scene(uint32_t width, uint32_t height, int n_spheres, int n_lights, uint64_t seed = 42)
{
    SyntheticScene s;
    s.width = width;
    s.height = height;
    Pcg32 rng(seed);
    const float ar = (float)width / (float)height;
    const float scale = std::sqrt(1024.0f / (float)n_spheres);
    for (int k = 0; k < n_spheres; ++k) {
        const float z = -(4.0f + 26.0f * rng.u());
        const float az = std::fabs(z);
        float x = 2.0f * rng.u();
        x = x - 1.0f; x = x * 0.55f; x = x * az; x = x * ar;
        float y = 2.0f * rng.u();
        y = y - 1.0f; y = y * 0.55f; y = y * az;
        float r = 0.02f * az;
        const float h = 0.5f + rng.u();
        r = r * h; r = r * scale;
        s.cx.push_back(x); s.cy.push_back(y); s.cz.push_back(z); s.radius.push_back(r);
        float c;
        c = 0.8f * rng.u(); s.rr.push_back(0.2f + c);
        c = 0.8f * rng.u(); s.rg.push_back(0.2f + c);
        c = 0.8f * rng.u(); s.rb.push_back(0.2f + c);
    }
    for (int m = 0; m < n_lights; ++m) {
        float v;
        v = 12.0f * rng.u(); s.px.push_back(v - 6.0f);
        v = 12.0f * rng.u(); s.py.push_back(v - 6.0f);
        v = 12.0f * rng.u(); s.pz.push_back(-v);
        s.lr.push_back(1.0f); s.lg.push_back(1.0f); s.lb.push_back(1.0f);
        s.intensity.push_back(20.0f / (float)(n_lights > 0 ? n_lights : 1));
        s.lkind.push_back((uint8_t)RT_POINT);
    }
    return s;
}

// BASELINE.json configs
inline bool config_scene(const char* name, SyntheticScene& out)
{
    if (!std::strcmp(name, "C3")) { out = generate_scene(3840, 2160, 1024, 4); return true; }
    if (!std::strcmp(name, "C4")) { out = generate_scene(7680, 4320, 8192, 8); return true; }
    if (!std::strcmp(name, "C5")) { out = generate_scene(3840, 2160, 4096, 32); return true; }
    return false;
}

}  // namespace society
