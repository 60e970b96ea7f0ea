// scene_prep.h — host-side flattening/preprocessing of the scene for the gfx950 kernels (no device code).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rt_mi355x.h"

namespace rtamd {

struct f4 {
    float x, y, z, w;
};

// Spheres are padded to a multiple of this many entries so the filter loop never needs a bounds check.
#ifndef RT_PAD
#define RT_PAD 16
#endif
constexpr int kSphereBlockPad = RT_PAD;
constexpr int kShadowCluster = 16;  // entries per cluster of the shadow any-hit hierarchy (scene_prep.cpp: prepare_light_filters)
static_assert(kSphereBlockPad % kShadowCluster == 0, "n_pad must be a whole number of clusters");

struct PreparedScene {
    int32_t n = 0;       // elements (renderer.rs:37)
    int32_t n_pad = 0;   // n rounded up to kSphereBlockPad
    int32_t n_lights = 0;
    std::vector<f4> filt;        // n_pad: (cx', cy', cz', 0) — conservative primary-ray filter d.c' - 1 >= 0 ; (0,0,0,0) = never passes
    std::vector<f4> sph;         // n_pad: (cx, cy, cz, k) with k = dot3(c,c) - r*r in the reference's order (sphere.rs:29 with origin 0)
    std::vector<f4> sph_r;       // n_pad: (cx, cy, cz, radius) for general-origin rays (bounces, shadow rays)
    std::vector<f4> gen;         // n_pad: (c'x, c'y, c'z, Kc), c' = c - ball centre, Kc = |c'|^2 (1 - 64u) - rho: general-origin filter
    std::vector<f4> refl;        // n_pad: Material.reflectiveness (element.rs:18), w unused
    std::vector<int32_t> always; // indices of spheres the filter cannot cover (origin inside / on / absurd values): tested exactly for every pixel
    std::vector<uint8_t> cls;    // n: 1 = filterable sphere, 0 = sphere always tested exactly, 2 = plane
    std::vector<f4> pln;         // n_pad: (nx, ny, nz, 1) for RT_PLANE elements, 0 otherwise
    std::vector<int32_t> planes; // indices of the RT_PLANE elements: tested exactly for every ray (no filter)
    double ball_c[3] = {0, 0, 0}, ball_r = 0;  // the same centre (as doubles of the f32 values) and the full radius Rs
    float ball[4] = {0, 0, 0, 0}; // centre and squared radius of the ball the shadow filters were derived for (hit points outside it
                                  // — possible only on planes — take the exact path)
    std::vector<f4> lpos;        // n_lights: (px, py, pz, intensity)
    std::vector<f4> lcol;        // n_lights: (cr, cg, cb, kind as float: 0 = point, 1 = ambient)
    // RT_SHADOWS: per light the shadow rays share the end point lp, so seen from the light they are "primary rays" and the
    // 3-FMA filter applies with sphere centres relative to lp and an inflated radius (scene_prep.cpp: prepare_light_filters)
    std::vector<int32_t> order;        // n_pad: element at spatial position j (Morton order of the sphere centres; planes last), -1 = padding
    std::vector<f4> sph_o;             // n_pad: sph_r in spatial order (the exact-test operands of shadow candidates)
    std::vector<f4> gen_o;             // n_pad: gen in spatial order (what the bounce trace lists and walks)
    std::vector<f4> gcl;               // n_pad / 16: per cluster of 16 consecutive positions (centre c' = c - ball centre, radius of the centres +
                                       // largest member radius (1 + 1e-4), rounded up; +inf = a member outside the argument; -1 = no member)
    std::vector<int32_t> rank;         // n_pad: spatial position of element i (the sort key of the hit records)
    std::vector<f4> lfilt;             // n_lights * n_pad, in SPATIAL order: (c''x, c''y, c''z, P2 of the member test); all zero = no entry
    std::vector<f4> lclA, lclB;        // n_lights * n_pad / 16: per cluster of 16 consecutive entries (axis x, y, z, dmin), (cos, sin of the half-angle, 0, 0)
    std::vector<int32_t> lalways;      // per light, indices of spheres its filter cannot cover (concatenated)
    std::vector<int32_t> lalways_off;  // n_lights + 1 offsets into lalways
    // RT_CULL: the same cone hierarchy for the PRIMARY rays (common origin 0), over the primary filter's own constants, in an order of its
    // own: the filterable spheres sorted by the DIRECTION they are seen in from the origin (scene_prep.cpp: prepare_primary_cull)
    std::vector<int32_t> porder;       // n_pad: element at position j of that order, -1 = no entry
    std::vector<f4> pfilt_o;           // n_pad: (cx', cy', cz', P2 of the member test) = filt in that order; all zero = no entry
    std::vector<f4> pclA, pclB;        // n_pad / 16: per cluster of 16 consecutive positions (cone axis x, y, z, 0), (cos, sin of the half-angle, 0, 0)
};

// Returns empty string on success, else an error text. Validates kinds and pointers.
std::string prepare_scene(const rt_objects* o, const rt_lights* l, PreparedScene& out);

// Filter constants only (used by rt_debug_prepare_spheres and prepare_scene).
void prepare_filter(int32_t n, const float* cx, const float* cy, const float* cz, const float* radius, f4* filt, uint8_t* cls, float* k_out);

// Camera::transform (display.rs:30-41)
void camera_transform(float fov, float z_near, float z_far, float aspect, float m[16]);
// Transform::apply (math/mod.rs:40-43) with Versor::apply (quat.rs:159-166): host-side pre-transform of object centres
void transform_apply(const float position[3], const float versor[4], const float scale[3], const float v[3], float out[3]);

}  // namespace rtamd
