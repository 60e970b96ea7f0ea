// scene_prep.cpp — host-side scene flattening for the gfx950 render kernels.
// Compiled with -ffp-contract=off: `k` below must carry exactly the bits the reference computes.
#include "scene_prep.h"

#include <cmath>
#include <cstring>

namespace rtamd {

// math/vec3.rs:118-155 — (xx + yy) + (zz + 0), every product rounded, never fused.
static inline float dot3_ref(float ax, float ay, float az, float bx, float by, float bz)
{
    float p0 = ax * bx, p1 = ay * by, p2 = az * bz;
    float lo = p0 + p1;
    float hi = p2 + 0.0f;
    return lo + hi;
}

/*
 * Primary-ray filter (origin == 0, renderer.rs:43).  For a sphere with k = |c|^2 - r^2 > 0 (origin strictly outside) the
 * reference reports a hit only through the `Greater` branch of sphere.rs:34-41, which needs  fl(a*a) >= k  with
 * a = dot3(d, c) > 0, i.e.  a >= sqrt(k) (1 - u),  u = 2^-24.  The kernel evaluates
 *        f = fma(dx, cx', fma(dy, cy', fma(dz, cz', -1)))      with  c' = c * (1 + mu) / sqrt(k)
 * and treats f >= 0 as "candidate".  With M = |cx|+|cy|+|cz| the rounding of a (<= 3.01 u M), of c' (u) and of the three
 * FMAs (<= 4.04 u (1 + M/sqrt(k))) is covered by  mu = 16 u (1 + M/sqrt(k)), so every reference hit is a candidate
 * (derivation in DESIGN.md).  Candidates are then re-tested with the reference's exact unfused arithmetic, so the filter
 * only ever costs time, never changes a pixel.  Spheres the argument does not cover (k <= 0, M/sqrt(k) > 2^10,
 * non-finite or absurdly scaled values) are classed "always": tested exactly for every pixel.
 */
void prepare_filter(int32_t n, const float* cx, const float* cy, const float* cz, const float* radius, f4* filt, uint8_t* cls, float* k_out)
{
    const double u = std::ldexp(1.0, -24);
    for (int32_t i = 0; i < n; ++i) {
        const float x = cx[i], y = cy[i], z = cz[i], r = radius[i];
        const float k = dot3_ref(x, y, z, x, y, z) - (r * r);  // sphere.rs:29 with dist = -c
        if (k_out) k_out[i] = k;
        bool ok = std::isfinite(x) && std::isfinite(y) && std::isfinite(z) && std::isfinite(r) && std::isfinite(k) && k > 0.0f;
        double M = std::fabs((double)x) + std::fabs((double)y) + std::fabs((double)z);
        double s = ok ? std::sqrt((double)k) : 1.0;
        if (ok && !(M >= 1e-12 && M <= 1e12)) ok = false;
        if (ok && !(M / s <= 1024.0)) ok = false;
        if (ok) {
            const double mu = 16.0 * u * (1.0 + M / s);
            const double g = (1.0 + mu) / s;
            filt[i] = f4{(float)(x * g), (float)(y * g), (float)(z * g), 0.0f};
            if (cls) cls[i] = 1;
        } else {
            filt[i] = f4{0.0f, 0.0f, 0.0f, 0.0f};  // f == -1: never a candidate
            if (cls) cls[i] = 0;
        }
    }
}

std::string prepare_scene(const rt_objects* o, const rt_lights* l, PreparedScene& out)
{
    if (!o || !l) return "objects/lights is null";
    if (o->n < 0 || l->n < 0) return "negative count";
    if (o->n > 0 && (!o->cx || !o->cy || !o->cz || !o->radius || !o->refl_r || !o->refl_g || !o->refl_b)) return "objects: null array";
    if (l->n > 0 && (!l->cr || !l->cg || !l->cb)) return "lights: null colour array";
    for (int32_t i = 0; i < o->n; ++i)
        if (o->kind && o->kind[i] != RT_SPHERE) return "objects: unknown kind (only RT_SPHERE exists, object/sphere.rs)";
    for (int32_t i = 0; i < l->n; ++i) {
        uint8_t k = l->kind ? l->kind[i] : (uint8_t)RT_POINT;
        if (k != RT_POINT && k != RT_AMBIENT) return "lights: unknown kind";
        if (k == RT_POINT && (!l->px || !l->py || !l->pz || !l->intensity)) return "lights: null position/intensity array for a point light";
    }
    out = PreparedScene();
    out.n = o->n;
    out.n_pad = ((o->n + kSphereBlockPad - 1) / kSphereBlockPad) * kSphereBlockPad;
    if (out.n_pad == 0) out.n_pad = kSphereBlockPad;
    out.n_lights = l->n;
    // padding entries: centre 0, k = +inf => delta = -inf => sqrt = NaN => miss (sphere.rs:34); radius NaN for the general test
    const float inf = INFINITY;
    out.filt.assign(out.n_pad, f4{0, 0, 0, 0});
    out.sph.assign(out.n_pad, f4{0, 0, 0, inf});
    out.sph_r.assign(out.n_pad, f4{0, 0, 0, NAN});
    out.gen.assign(out.n_pad, f4{0, 0, 0, inf});  // -rho = +inf: W = +inf, g = -inf, never a candidate
    out.refl.assign(out.n_pad, f4{0, 0, 0, 0});
    out.cls.assign(o->n, 0);
    std::vector<float> k(o->n);
    prepare_filter(o->n, o->cx, o->cy, o->cz, o->radius, out.filt.data(), out.cls.data(), k.data());
    for (int32_t i = 0; i < o->n; ++i) {
        out.sph[i] = f4{o->cx[i], o->cy[i], o->cz[i], k[i]};
        out.sph_r[i] = f4{o->cx[i], o->cy[i], o->cz[i], o->radius[i]};
        {
            // general-origin filter (render_kernels.hip): rho = fl(r*r) (1+eta)/(1-eta), eta = 2^-17, rounded up
            const double eta = std::ldexp(1.0, -17);
            const float r2 = o->radius[i] * o->radius[i];  // the reference's own rounded r*r (sphere.rs:29)
            float rho = (float)((double)r2 * (1.0 + eta) / (1.0 - eta));
            rho = std::nextafter(rho, INFINITY);
            out.gen[i] = f4{o->cx[i], o->cy[i], o->cz[i], -rho};
        }
        out.refl[i] = f4{o->refl_r[i], o->refl_g[i], o->refl_b[i], 0.0f};
        if (!out.cls[i]) out.always.push_back(i);
    }
    out.lpos.resize(l->n);
    out.lcol.resize(l->n);
    for (int32_t i = 0; i < l->n; ++i) {
        uint8_t kd = l->kind ? l->kind[i] : (uint8_t)RT_POINT;
        if (kd == RT_POINT)
            out.lpos[i] = f4{l->px[i], l->py[i], l->pz[i], l->intensity[i]};
        else
            out.lpos[i] = f4{0, 0, 0, 0};
        out.lcol[i] = f4{l->cr[i], l->cg[i], l->cb[i], (float)kd};
    }
    return std::string();
}

void camera_transform(float fov, float z_near, float z_far, float aspect, float m[16])
{
    // display.rs:30-41
    float yy = 1.0f / std::tan(fov / 2.0f);
    float zm = z_far - z_near;
    float zp = z_far + z_near;
    std::memset(m, 0, 16 * sizeof(float));
    m[0] = yy / aspect;
    m[5] = yy;
    m[10] = -zp / zm;
    m[11] = ((-2.0f * z_far) * z_near) / zm;
    m[14] = -1.0f;
}

}  // namespace rtamd
