// scene_prep.cpp — host-side scene flattening for the gfx950 render kernels.
// Compiled with -ffp-contract=off: `k` below must carry exactly the bits the reference computes.
#include "scene_prep.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>

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

// Bounding ball of everything a ray can start from or end at: the sphere surfaces and the camera (0,0,0); the lights are added per
// light (prepare_light_filters).  Planes are unbounded: they do not shape the ball; hit points outside it take exact paths in the kernel.
// Its centre is also the point the general-origin filter is expanded around (out.gen).
static void scene_ball(const rt_objects* o, PreparedScene& out)
{
    const int32_t n = o->n;
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    auto fin = [&](int32_t i) {
        return !(o->kind && o->kind[i] == RT_PLANE) && std::isfinite(o->cx[i]) && std::isfinite(o->cy[i]) && std::isfinite(o->cz[i]) && std::isfinite(o->radius[i]);
    };
    for (int32_t i = 0; i < n; ++i) {
        if (!fin(i)) continue;
        const double v[3] = {o->cx[i], o->cy[i], o->cz[i]};
        for (int a = 0; a < 3; ++a) {
            if (v[a] < lo[a]) lo[a] = v[a];
            if (v[a] > hi[a]) hi[a] = v[a];
        }
    }
    for (int a = 0; a < 3; ++a) {
        out.ball[a] = (float)(0.5 * (lo[a] + hi[a]));
        out.ball_c[a] = out.ball[a];  // the centre exactly as the kernel sees it
    }
    auto distC = [&](double x, double y, double z) {
        return std::sqrt((x - out.ball_c[0]) * (x - out.ball_c[0]) + (y - out.ball_c[1]) * (y - out.ball_c[1]) + (z - out.ball_c[2]) * (z - out.ball_c[2]));
    };
    double Rs = distC(0, 0, 0);
    for (int32_t i = 0; i < n; ++i)
        if (fin(i)) Rs = std::fmax(Rs, distC(o->cx[i], o->cy[i], o->cz[i]) + std::fabs((double)o->radius[i]));
    out.ball_r = Rs;
    // hit points within 0.999 Rs of the centre (in f32 arithmetic) are certainly inside the ball the shadow margins assume
    out.ball[3] = (float)(0.998 * Rs * Rs);
}

/*
 * Shadow-ray filter (RT_SHADOWS extension).  A shadow ray runs from a hit point h toward the light at lp with the float
 * direction u = (lp - h)/dist; the kernel tests the reversed line {lp + t w}, w = -u, which has a COMMON origin for all rays
 * of one light, so the primary-ray filter  f = w.c'' - 1 >= 0  applies with q' = c - lp.  For every (ray, sphere) the
 * reference's exact test could accept, the filter must pass:
 *   - the reference evaluates delta from q = h - c in f32: its delta >= 0 only guarantees dist(c, line)^2 <= r^2 + 55 u (|q|^2 + r^2);
 *   - the reversed line is parallel to the reference's, offset by the rounding residual of u (<= 6 u dist);
 *   - |w| = 1 +- 3u.
 * With S >= every |q| and every dist in the scene this is covered by testing against the inflated radius
 *   R^2 = r^2 + 128 u (S^2 + r^2)      and      mu = 20 u (1 + M/sqrt(k)),  k = |q'|^2 - R^2.
 * Spheres with k <= 0 (the light is inside / within the noise band of the sphere), M/sqrt(k) > 2^10 or non-finite data are
 * listed per light and tested exactly for every ray.  Candidates always go through the reference's exact test.
 *
 * Any-hit hierarchy on top of the filter (round 3).  The constants are stored in the SPATIAL ORDER of the spheres
 * (out.order: Morton order of the centres), 16 consecutive entries form a CLUSTER, and the kernel bounds the reversed directions w of
 * the 256 rays a wave holds by a cone (unit axis a, half-angle theta <= 60 deg).  A ray inside that cone can only be a candidate of
 * sphere s (f >= 0 in the kernel's FMA arithmetic, whose error is <= 4.04 u (1 + sqrt(3) |c''|)) if, for |c''| <= 4,
 *        w.c'' >= 1 - 33u   =>   angle(a, c'') <= theta + theta_s ,  cos(theta_s) = P1 / |c''| ,  P1 = 1 - 64u
 *   <=>  a.c'' >= cos(theta) P1 - sin(theta) P2 ,   P2 = sqrt(|c''|^2 - P1^2)          (the MEMBER test; P2 rides in lfilt.w)
 * and a cluster with axis e and half-angle theta_c >= angle(e, c''_s) + theta_s for all its members s can only hold a candidate if
 *        a.e >= cos(theta) cos(theta_c) - sin(theta) sin(theta_c)                      (the CLUSTER test; lclA / lclB).
 * Both are evaluated with a few-ulp slack in the kernel; spheres / clusters outside the argument (|c''| > 4, theta_c > 80 deg) always
 * pass.  Independently of direction, a blocker the reference can report (RT_SHADOW_EPS < t < dist) has
 *        |c - lp| <= dist + r + sqrt(55 u (S^2 + r^2)) ,
 * so a cluster whose members are all farther from the light than every ray of the wave is long (dmin = min(|c - lp| - r - 6e-3 S))
 * is skipped as well.  Neither test ever removes a sphere the per-ray filter would pass: the output is unchanged by construction.
 */
static void prepare_light_filters(const rt_objects* o, const rt_lights* l, PreparedScene& out)
{
    const double u = std::ldexp(1.0, -24);
    const int32_t n = o->n, nl = l->n, np = out.n_pad, ncl = np / kShadowCluster;
    out.lfilt.assign((size_t)nl * np, f4{0, 0, 0, 0});
    out.lclA.assign((size_t)nl * ncl, f4{0, 0, 0, INFINITY});  // dmin = +inf: an empty cluster never passes
    out.lclB.assign((size_t)nl * ncl, f4{2.0f, 0, 0, 0});
    out.lalways.clear();
    out.lalways_off.assign(nl + 1, 0);
    std::vector<char> fin(n, 0);
    auto is_plane = [&](int32_t i) { return o->kind && o->kind[i] == RT_PLANE; };
    for (int32_t i = 0; i < n; ++i)
        fin[i] = !is_plane(i) && std::isfinite(o->cx[i]) && std::isfinite(o->cy[i]) && std::isfinite(o->cz[i]) && std::isfinite(o->radius[i]);
    const double C[3] = {out.ball_c[0], out.ball_c[1], out.ball_c[2]}, Rs = out.ball_r;
    auto distC = [&](double x, double y, double z) { return std::sqrt((x - C[0]) * (x - C[0]) + (y - C[1]) * (y - C[1]) + (z - C[2]) * (z - C[2])); };
    const double P1 = 1.0 - 64.0 * u;
    std::vector<double> th_s(kShadowCluster), ex(kShadowCluster), ey(kShadowCluster), ez(kShadowCluster);
    for (int32_t m = 0; m < nl; ++m) {
        const uint8_t kind = l->kind ? l->kind[m] : (uint8_t)RT_POINT;
        out.lalways_off[m] = (int32_t)out.lalways.size();
        if (kind != RT_POINT) continue;
        const double lx = l->px[m], ly = l->py[m], lz = l->pz[m];
        const bool lfin = std::isfinite(lx) && std::isfinite(ly) && std::isfinite(lz);
        const double Sl = lfin ? distC(lx, ly, lz) + Rs : 0.0;  // >= every |lp - h|
        for (int32_t cl = 0; cl < ncl; ++cl) {
            int members = 0;
            bool open_dir = false;  // some member is outside the cone argument: the cluster passes every direction test
            double dmin = INFINITY, sx = 0, sy = 0, sz = 0;
            for (int32_t jj = 0; jj < kShadowCluster; ++jj) {
                const int32_t j = cl * kShadowCluster + jj;
                const int32_t i = out.order[j];
                if (i < 0 || is_plane(i)) continue;  // lfilt stays 0 (never a candidate); planes are tested exactly through out.planes
                bool ok = fin[i] && lfin;
                double qx = 0, qy = 0, qz = 0, s = 1, M = 0, dl = 0, S = 0;
                if (ok) {
                    qx = o->cx[i] - lx; qy = o->cy[i] - ly; qz = o->cz[i] - lz;
                    const double r2 = (double)o->radius[i] * (double)o->radius[i];
                    S = 1.01 * std::fmax(distC(o->cx[i], o->cy[i], o->cz[i]) + Rs, Sl);  // >= every |h - c_i| and every dist
                    const double R2 = r2 + 128.0 * u * (S * S + r2);
                    dl = std::sqrt(qx * qx + qy * qy + qz * qz);
                    const double k = qx * qx + qy * qy + qz * qz - R2;
                    M = std::fabs(qx) + std::fabs(qy) + std::fabs(qz);
                    ok = std::isfinite(k) && k > 0.0 && M >= 1e-12 && M <= 1e12;
                    if (ok) {
                        s = std::sqrt(k);
                        ok = M / s <= 1024.0;
                    }
                }
                if (!ok) {
                    out.lalways.push_back(i);
                    continue;
                }
                const double mu = 20.0 * u * (1.0 + M / s);
                const double g = (1.0 + mu) / s;
                f4 c2 = f4{(float)(qx * g), (float)(qy * g), (float)(qz * g), 0.0f};
                if (c2.x == 0.0f && c2.y == 0.0f && c2.z == 0.0f) {  // (cannot happen for M/s >= 1; all-zero constants mean "no entry" to the kernel)
                    out.lalways.push_back(i);
                    continue;
                }
                // member test constants from the f32 values the kernel multiplies with
                const double nx = c2.x, ny = c2.y, nz = c2.z, nn = std::sqrt(nx * nx + ny * ny + nz * nz);
                double p2 = nn > P1 ? std::sqrt(nn * nn - P1 * P1) : 0.0;
                c2.w = std::nextafter((float)p2, INFINITY);
                if (!(nn <= 4.0)) {  // the light is close to the surface: theta_s > 75 deg, outside the cone argument
                    c2.w = 1.0e30f;  // rhs ~ -1e30 sin(theta) with sin(theta) >= 1e-6 in the kernel: the member test always passes
                    open_dir = true;
                }
                out.lfilt[(size_t)m * np + j] = c2;
                th_s[members] = nn > P1 ? std::acos(P1 / nn) : 0.0;
                ex[members] = nx / nn; ey[members] = ny / nn; ez[members] = nz / nn;
                sx += ex[members]; sy += ey[members]; sz += ez[members];
                ++members;
                dmin = std::fmin(dmin, dl - std::fabs((double)o->radius[i]) - 6e-3 * S);
            }
            if (!members) continue;
            f4 A = f4{0, 0, 0, std::nextafter((float)dmin, -INFINITY)}, B = f4{-2.0f, 0, 0, 0};  // cos = -2: passes every direction test
            const double sn = std::sqrt(sx * sx + sy * sy + sz * sz);
            if (!open_dir && sn > 1e-6) {
                const float ax = (float)(sx / sn), ay = (float)(sy / sn), az = (float)(sz / sn);  // the axis as the kernel sees it
                const double an = std::sqrt((double)ax * ax + (double)ay * ay + (double)az * az);
                double th = 0.0;
                for (int k = 0; k < members; ++k) {
                    double c = (ex[k] * ax + ey[k] * ay + ez[k] * az) / an;
                    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
                    th = std::fmax(th, std::acos(c) + th_s[k]);
                }
                th += 1e-5;
                if (th <= 80.0 * 3.14159265358979323846 / 180.0) {
                    A.x = ax; A.y = ay; A.z = az;
                    B.x = std::nextafter((float)std::cos(th), -INFINITY);
                    B.y = std::nextafter((float)std::sin(th), INFINITY);
                }
            }
            out.lclA[(size_t)m * ncl + cl] = A;
            out.lclB[(size_t)m * ncl + cl] = B;
        }
    }
    out.lalways_off[nl] = (int32_t)out.lalways.size();
}

/*
 * RT_CULL: the cone hierarchy of the shadow rays (above) applied to the PRIMARY rays.  All primary rays start at the origin
 * (renderer.rs:43), so a wave's 16 x 16 pixel tile is a narrow cone of directions d, and the primary filter  f = d.c' - 1 >= 0  is the
 * shadow filter with the "light" at the origin.  The member and cluster tests are the ones derived above with c'' = c' (the f32
 * constants of prepare_filter, |c'| <= 4 or always open): a ray inside the cone (a, theta) can be a candidate of sphere s only if
 * a.c' >= cos(theta) P1 - sin(theta) P2.  There is no length test (primary rays have no far end).  The filterable spheres are put in an
 * order of their own — Morton order of the octahedral map of their direction c / |c| — so that 16 consecutive entries are neighbours ON
 * THE SCREEN of any camera at the origin, whatever its matrix; spheres on the always-list and planes have no entry (the kernel tests
 * them exactly for every pixel, as without RT_CULL).  Neither test drops a sphere the per-pixel filter would pass: same image by construction.
 */
static void prepare_primary_cull(PreparedScene& out)
{
    const double u = std::ldexp(1.0, -24), P1 = 1.0 - 64.0 * u;
    const int32_t n = out.n, np = out.n_pad, ncl = np / kShadowCluster;
    out.porder.assign(np, -1);
    out.pfilt_o.assign(np, f4{0, 0, 0, 0});
    out.pclA.assign(ncl, f4{0, 0, 0, 0});
    out.pclB.assign(ncl, f4{2.0f, 0, 0, 0});  // cos = 2: an empty cluster fails every direction test
    std::vector<std::pair<uint64_t, int32_t>> keyed;
    auto spread = [](uint64_t v) {  // 16 bits -> every second bit
        uint64_t r = 0;
        for (int b = 0; b < 16; ++b) r |= ((v >> b) & 1ull) << (2 * b);
        return r;
    };
    for (int32_t i = 0; i < n; ++i) {
        if (out.cls[i] != 1) continue;
        const f4 c = out.filt[i];
        const double nn = std::sqrt((double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z);
        if (!(nn > 0.0) || !std::isfinite(nn)) continue;  // (cannot happen for a filterable sphere: |c'| >= 1)
        // octahedral map of the direction onto [0, 1]^2
        const double l1 = std::fabs((double)c.x) + std::fabs((double)c.y) + std::fabs((double)c.z);
        double ox = c.x / l1, oy = c.y / l1;
        if (c.z > 0.0f) {  // fold the far hemisphere over (the default camera looks down -z: the near one stays contiguous)
            const double fx = (1.0 - std::fabs(oy)) * (ox >= 0 ? 1.0 : -1.0), fy = (1.0 - std::fabs(ox)) * (oy >= 0 ? 1.0 : -1.0);
            ox = fx; oy = fy;
        }
        auto q16 = [](double t) {
            t = 0.5 * (t + 1.0);
            t = t < 0 ? 0 : (t > 1 ? 1 : t);
            return (uint64_t)(t * 65535.0);
        };
        keyed.emplace_back((spread(q16(ox)) | (spread(q16(oy)) << 1)) << 22 | (uint64_t)i, i);
    }
    std::sort(keyed.begin(), keyed.end());
    std::vector<double> th_s(kShadowCluster), ex(kShadowCluster), ey(kShadowCluster), ez(kShadowCluster);
    for (int32_t cl = 0; cl < ncl; ++cl) {
        int members = 0;
        bool open_dir = false;
        double sx = 0, sy = 0, sz = 0;
        for (int32_t jj = 0; jj < kShadowCluster; ++jj) {
            const int32_t j = cl * kShadowCluster + jj;
            if ((size_t)j >= keyed.size()) break;
            const int32_t i = keyed[j].second;
            f4 c = out.filt[i];
            const double nx = c.x, ny = c.y, nz = c.z, nn = std::sqrt(nx * nx + ny * ny + nz * nz);
            const double p2 = nn > P1 ? std::sqrt(nn * nn - P1 * P1) : 0.0;
            c.w = std::nextafter((float)p2, INFINITY);
            if (!(nn <= 4.0)) {  // the origin is close to the surface: theta_s > 75 deg, outside the cone argument
                c.w = 1.0e30f;
                open_dir = true;
            }
            out.porder[j] = i;
            out.pfilt_o[j] = c;
            th_s[members] = nn > P1 ? std::acos(P1 / nn) : 0.0;
            ex[members] = nx / nn; ey[members] = ny / nn; ez[members] = nz / nn;
            sx += ex[members]; sy += ey[members]; sz += ez[members];
            ++members;
        }
        if (!members) continue;
        f4 A = f4{0, 0, 0, 0}, B = f4{-2.0f, 0, 0, 0};  // cos = -2: passes every direction test
        const double sn = std::sqrt(sx * sx + sy * sy + sz * sz);
        if (!open_dir && sn > 1e-6) {
            const float ax = (float)(sx / sn), ay = (float)(sy / sn), az = (float)(sz / sn);  // the axis as the kernel sees it
            const double an = std::sqrt((double)ax * ax + (double)ay * ay + (double)az * az);
            double th = 0.0;
            for (int k = 0; k < members; ++k) {
                double c = (ex[k] * ax + ey[k] * ay + ez[k] * az) / an;
                c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
                th = std::fmax(th, std::acos(c) + th_s[k]);
            }
            th += 1e-5;
            if (th <= 80.0 * 3.14159265358979323846 / 180.0) {
                A.x = ax; A.y = ay; A.z = az;
                B.x = std::nextafter((float)std::cos(th), -INFINITY);
                B.y = std::nextafter((float)std::sin(th), INFINITY);
            }
        }
        out.pclA[cl] = A;
        out.pclB[cl] = B;
    }
}

// Spatial order of the elements (Morton code of the sphere centres, 10 bits per axis over their bounding box; planes and non-finite
// spheres last, in index order): out.order[j] = element at position j (-1 = padding), out.rank[i] = position of element i.
// Consecutive positions are neighbours in space, so (i) 16 consecutive entries make a tight cluster for the shadow any-hit hierarchy
// and (ii) hit records sorted by the rank of their element make waves whose shadow rays toward one light form a narrow cone.
static void spatial_order(const rt_objects* o, PreparedScene& out)
{
    const int32_t n = o->n;
    std::vector<uint64_t> key(n);
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto fin = [&](int32_t i) {
        return !(o->kind && o->kind[i] == RT_PLANE) && std::isfinite(o->cx[i]) && std::isfinite(o->cy[i]) && std::isfinite(o->cz[i]);
    };
    for (int32_t i = 0; i < n; ++i) {
        if (!fin(i)) continue;
        const double v[3] = {o->cx[i], o->cy[i], o->cz[i]};
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::fmin(lo[a], v[a]);
            hi[a] = std::fmax(hi[a], v[a]);
        }
    }
    auto spread = [](uint64_t v) {  // 10 bits -> every third bit
        uint64_t r = 0;
        for (int b = 0; b < 10; ++b) r |= ((v >> b) & 1ull) << (3 * b);
        return r;
    };
    for (int32_t i = 0; i < n; ++i) {
        uint64_t code = 1ull << 40;  // after every finite sphere
        if (fin(i)) {
            const double v[3] = {o->cx[i], o->cy[i], o->cz[i]};
            uint64_t q[3];
            for (int a = 0; a < 3; ++a) {
                const double ext = hi[a] - lo[a];
                double t = ext > 0 ? (v[a] - lo[a]) / ext : 0.0;
                t = t < 0 ? 0 : (t > 1 ? 1 : t);
                q[a] = (uint64_t)(t * 1023.0);
            }
            code = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
        }
        key[i] = (code << 22) | (uint64_t)i;  // index as tie-break: a total, deterministic order (n < 2^22 is checked by the caller)
    }
    std::vector<int32_t> idx(n);
    for (int32_t i = 0; i < n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
    out.order.assign(out.n_pad, -1);
    out.rank.assign(out.n_pad, 0);
    for (int32_t j = 0; j < n; ++j) {
        out.order[j] = idx[j];
        out.rank[idx[j]] = j;
    }
}

std::string prepare_scene(const rt_objects* o, const rt_lights* l, PreparedScene& out)
{
    if (!o || !l) return "objects/lights is null";
    if (o->n < 0 || l->n < 0) return "negative count";
    if (o->n >= (1 << 22)) return "more than 2^22 - 1 elements";
    if (o->n > 0 && (!o->cx || !o->cy || !o->cz || !o->radius || !o->refl_r || !o->refl_g || !o->refl_b)) return "objects: null array";
    if (l->n > 0 && (!l->cr || !l->cg || !l->cb)) return "lights: null colour array";
    for (int32_t i = 0; i < o->n; ++i) {
        if (!o->kind || o->kind[i] == RT_SPHERE) continue;
        if (o->kind[i] != RT_PLANE) return "objects: unknown kind (RT_SPHERE = object/sphere.rs, RT_PLANE = extension)";
        if (!o->nx || !o->ny || !o->nz) return "objects: null normal array for a plane";
    }
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
    out.gen.assign(out.n_pad, f4{0, 0, 0, inf});  // Kc = +inf: W = +inf, g = -inf, never a candidate
    out.refl.assign(out.n_pad, f4{0, 0, 0, 0});
    out.pln.assign(out.n_pad, f4{0, 0, 0, 0});
    out.cls.assign(o->n, 0);
    scene_ball(o, out);
    std::vector<float> k(o->n);
    prepare_filter(o->n, o->cx, o->cy, o->cz, o->radius, out.filt.data(), out.cls.data(), k.data());
    for (int32_t i = 0; i < o->n; ++i) {
        if (o->kind && o->kind[i] == RT_PLANE) {
            // a plane never passes a sphere filter and never passes a sphere test (k = +inf, radius NaN, -rho = +inf: see the padding)
            out.filt[i] = f4{0, 0, 0, 0};
            out.cls[i] = 2;
            out.sph[i] = f4{o->cx[i], o->cy[i], o->cz[i], inf};
            out.sph_r[i] = f4{o->cx[i], o->cy[i], o->cz[i], NAN};
            out.pln[i] = f4{o->nx[i], o->ny[i], o->nz[i], 1.0f};
            out.refl[i] = f4{o->refl_r[i], o->refl_g[i], o->refl_b[i], 0.0f};
            out.planes.push_back(i);
            continue;
        }
        out.sph[i] = f4{o->cx[i], o->cy[i], o->cz[i], k[i]};
        out.sph_r[i] = f4{o->cx[i], o->cy[i], o->cz[i], o->radius[i]};
        {
            // general-origin filter (render_kernels.hip: general_filter): rho = fl(r*r) (1+eta)/(1-eta), eta = 2^-17, rounded up;
            // c' = c - B as f32; Kc = |c'|^2 (1 - 64u) - rho, rounded DOWN (a smaller W can only add candidates).  Non-finite data
            // gives Kc = NaN or +-inf: g is then NaN (never >= 0) exactly where the reference's own arithmetic yields NaN (a miss).
            const double eta = std::ldexp(1.0, -17), u64 = std::ldexp(1.0, -18);
            const float r2 = o->radius[i] * o->radius[i];  // the reference's own rounded r*r (sphere.rs:29)
            float rho = (float)((double)r2 * (1.0 + eta) / (1.0 - eta));
            rho = std::nextafter(rho, INFINITY);
            const float gx = (float)((double)o->cx[i] - out.ball_c[0]), gy = (float)((double)o->cy[i] - out.ball_c[1]), gz = (float)((double)o->cz[i] - out.ball_c[2]);
            const double c2 = (double)gx * gx + (double)gy * gy + (double)gz * gz;
            float kc = (float)(c2 * (1.0 - u64) - (double)rho);
            kc = std::nextafter(kc, -INFINITY);
            // absurd scales (squares near the f32 range): always a candidate, the exact test decides
            if (std::fabs(gx) > 1e15f || std::fabs(gy) > 1e15f || std::fabs(gz) > 1e15f || std::fabs(o->radius[i]) > 1e15f) kc = -INFINITY;
            out.gen[i] = f4{gx, gy, gz, kc};
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
    spatial_order(o, out);
    out.sph_o.assign(out.n_pad, f4{0, 0, 0, NAN});
    for (int32_t j = 0; j < out.n_pad; ++j)
        if (out.order[j] >= 0) out.sph_o[j] = out.sph_r[out.order[j]];
    // bounce trace (render_kernels.hip: trace_bundle): the general-origin filter constants in spatial order and, per cluster of 16
    // consecutive positions, a bounding sphere of the members (centres c' as the kernel sees them, in f32)
    out.gen_o.assign(out.n_pad, f4{0, 0, 0, inf});
    for (int32_t j = 0; j < out.n_pad; ++j)
        if (out.order[j] >= 0) out.gen_o[j] = out.gen[out.order[j]];
    const int32_t ncl = out.n_pad / kShadowCluster;
    out.gcl.assign(ncl, f4{0, 0, 0, -1.0f});
    for (int32_t cl = 0; cl < ncl; ++cl) {
        double sx = 0, sy = 0, sz = 0;
        int members = 0;
        bool open = false;
        for (int32_t jj = 0; jj < kShadowCluster; ++jj) {
            const int32_t j = cl * kShadowCluster + jj;
            const f4 g = out.gen_o[j];
            if (g.w == inf) continue;  // padding / plane: never a candidate
            const float r = out.sph_o[j].w;
            if (!(std::isfinite(g.x) && std::isfinite(g.y) && std::isfinite(g.z) && std::isfinite(g.w) && std::isfinite(r))) {
                open = true;  // non-finite data, or Kc = -inf (a sphere outside the filter's argument): the cluster is never skipped
                continue;
            }
            sx += g.x; sy += g.y; sz += g.z;
            ++members;
        }
        if (!members && !open) continue;
        f4 c = f4{0, 0, 0, inf};
        if (!open) {
            c.x = (float)(sx / members); c.y = (float)(sy / members); c.z = (float)(sz / members);
            double R = 0;
            for (int32_t jj = 0; jj < kShadowCluster; ++jj) {
                const int32_t j = cl * kShadowCluster + jj;
                const f4 g = out.gen_o[j];
                if (g.w == inf) continue;
                const double ex = (double)g.x - c.x, ey = (double)g.y - c.y, ez = (double)g.z - c.z;
                R = std::fmax(R, std::sqrt(ex * ex + ey * ey + ez * ez) + std::fabs((double)out.sph_o[j].w) * (1.0 + 1e-4));
            }
            c.w = std::nextafter((float)(R * (1.0 + 1e-6)), INFINITY);
        }
        out.gcl[cl] = c;
    }
    prepare_light_filters(o, l, out);
    prepare_primary_cull(out);
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

// quat.rs:201-226 with quaternions as (r, i, j, k): per lane (a0 b - t3) + ((t1 + t2) * (-1, 1, 1, 1)), every product and sum rounded
static void hamilton(const float a[4], const float b[4], float out[4])
{
    const float t1[4] = {a[1] * b[1], a[1] * b[0], a[2] * b[0], a[3] * b[0]};
    const float t2[4] = {a[2] * b[2], a[2] * b[3], a[3] * b[1], a[1] * b[2]};
    const float t3[4] = {a[3] * b[3], a[3] * b[2], a[1] * b[3], a[2] * b[1]};
    const float sign[4] = {-1.0f, 1.0f, 1.0f, 1.0f};
    for (int k = 0; k < 4; ++k) {
        const float t12 = t1[k] + t2[k];
        const float t12m = t12 * sign[k];
        const float t0 = a[0] * b[k];
        const float t03 = t0 - t3[k];
        out[k] = t03 + t12m;
    }
}

void transform_apply(const float position[3], const float versor[4], const float scale[3], const float v[3], float out[3])
{
    // math/mod.rs:40-43: position + rotation.apply(v).wide_mul(scale); quat.rs:159-166: (q * (0, v)) * q.inverse(), inverse = conjugate
    const float p[4] = {0.0f, v[0], v[1], v[2]};
    const float inv[4] = {versor[0] * 1.0f, versor[1] * -1.0f, versor[2] * -1.0f, versor[3] * -1.0f};
    float qp[4], res[4];
    hamilton(versor, p, qp);
    hamilton(qp, inv, res);
    for (int k = 0; k < 3; ++k) {
        const float scaled = res[k + 1] * scale[k];
        out[k] = position[k] + scaled;
    }
}

}  // namespace rtamd
