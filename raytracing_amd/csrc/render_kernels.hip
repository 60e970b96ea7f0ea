// render_kernels.hip — gfx950 (MI355X, CDNA4, wave64) kernels for the per-pixel render path of Aandreba/raytracing.
//
// What one launch computes (reference file:line under /root/reference/src):
//   display.rs:98-100   pixel (j,i) -> camera-space position            (per pixel, exact reference arithmetic)
//   renderer.rs:41-44   primary ray: origin 0, direction pos.unit()
//   renderer.rs:47-54   closest hit over all elements (strict <, lowest index wins ties)
//   object/sphere.rs:26-44  ray/sphere test
//   renderer.rs:56-66   hit point, sum of lights in list order (light/point.rs:25-30, light/ambient.rs:23-25), (x) reflectiveness
//   renderer.rs:67-71   bounce for depth > 1 (normal at the OLD origin — reference quirk), object/mod.rs:71-73
//   renderer.rs:75-76   255*c -> saturating u8
//
// Hot loop: every (pixel, sphere) pair is evaluated with a 3-FMA conservative filter  f = d.c' - 1  (scene_prep.cpp);
// a wave votes once per block of spheres and only candidate blocks run the reference's exact unfused test, so the
// output is bit-identical to the CPU reference while the inner loop is pure v_fma_f32 + v_max3_f32.
// The object list is staged once per workgroup in LDS in 1024-sphere chunks and read back as wave-uniform
// (broadcast) ds_read_b128.  Built with -ffp-contract=off: nothing is fused unless written as fmaf().
#include "render_kernels.h"

#include "rt_mi355x.h"

#pragma clang fp contract(off)

namespace rtamd {
namespace {

constexpr int kLX = 16;      // lanes of a wave along x
constexpr int kLY = 4;       // lanes of a wave along y
constexpr int kWaves = 4;    // waves per workgroup, side by side along x
constexpr int kChunk = 1024; // spheres staged in LDS per pass (16 KiB)
#ifndef RT_SHADOW_CAP
#define RT_SHADOW_CAP 256
#endif
#ifndef RT_SHADOW_B
#define RT_SHADOW_B 8
#endif
constexpr int kShadowB = RT_SHADOW_B;      // spheres per wave vote in the shadow loop
constexpr int kShadowCap = RT_SHADOW_CAP;  // entries of the per-wave sphere list of shadow_mask_kernel (16 B each in LDS)

// math/vec3.rs:118-155 — (xx + yy) + (zz + 0): every product and sum individually rounded
__device__ __forceinline__ float dot3_ref(float ax, float ay, float az, float bx, float by, float bz)
{
    float p0 = ax * bx, p1 = ay * by, p2 = az * bz;
    float lo = p0 + p1;
    float hi = p2 + 0.0f;
    return lo + hi;
}

// math/vec4.rs:84-121 — (p0 + p1) + (p2 + p3)
__device__ __forceinline__ float dot4_ref(const float* r, float v0, float v1, float v2, float v3)
{
    float p0 = r[0] * v0, p1 = r[1] * v1, p2 = r[2] * v2, p3 = r[3] * v3;
    float lo = p0 + p1;
    float hi = p2 + p3;
    return lo + hi;
}

// object/sphere.rs:34-41 given alpha (already negated, :31) and beta = sqrt(delta) (:32)
__device__ __forceinline__ bool root_select(float alpha, float beta, float& t)
{
    if (alpha != alpha || beta != beta) return false;  // partial_cmp -> None
    if (alpha == beta) { t = 0.0f; return true; }
    if (alpha > beta) { t = alpha - beta; return true; }
    float x = alpha + beta;
    if (x < 0.0f) return false;
    t = x;
    return true;
}

// object/sphere.rs:26-44 for origin == 0: dist = -c, so alpha(:31) = dot3(d, c) and dot3(dist,dist) - r*r = s.w (host, same bits)
__device__ __forceinline__ bool hit_primary_exact(float dx, float dy, float dz, float4 s, float& t)
{
    float a = dot3_ref(dx, dy, dz, s.x, s.y, s.z);
    float delta = (a * a) - s.w;
    float beta = sqrtf(delta);
    return root_select(a, beta, t);
}

// object/sphere.rs:26-44, general origin; s = (cx, cy, cz, radius)
__device__ __forceinline__ bool hit_general_exact(float ox, float oy, float oz, float dx, float dy, float dz, float4 s, float& t)
{
    float qx = ox - s.x, qy = oy - s.y, qz = oz - s.z;
    float alpha = dot3_ref(dx, dy, dz, qx, qy, qz);
    float delta = (alpha * alpha) - (dot3_ref(qx, qy, qz, qx, qy, qz) - (s.w * s.w));
    alpha = -alpha;
    float beta = sqrtf(delta);
    return root_select(alpha, beta, t);
}

// EXTENSION (RT_PLANE, no reference counterpart; semantics = the oracle's plane_is_hit_by): p0 = a point of the plane, n = unit normal
__device__ __forceinline__ bool hit_plane_exact(float ox, float oy, float oz, float dx, float dy, float dz, float4 p0, float4 n, float& t)
{
    const float denom = dot3_ref(dx, dy, dz, n.x, n.y, n.z);
    const float qx = p0.x - ox, qy = p0.y - oy, qz = p0.z - oz;
    const float num = dot3_ref(qx, qy, qz, n.x, n.y, n.z);
    const float x = num / denom;
    if (x != x || __builtin_isinf(x)) return false;  // parallel ray
    if (x < 0.0f) return false;
    t = x;
    return true;
}

// renderer.rs:49-53 made order-independent: lexicographic minimum of (t, index)
__device__ __forceinline__ void take_closest(bool hit, float t, int idx, float& best_t, int& best_i)
{
    if (hit && (best_i < 0 || t < best_t || (t == best_t && idx < best_i))) {
        best_t = t;
        best_i = idx;
    }
}

// renderer.rs:75-76 — Rust `as u8`: NaN -> 0, saturating, truncating
__device__ __forceinline__ uint32_t quantise(float c)
{
    float q = 255.0f * c;
    if (!(q > 0.0f)) return 0u;
    if (q >= 255.0f) return 255u;
    return (uint32_t)q;
}

// math/vec3.rs:195-197 — v / sqrt(dot(v,v)), true divisions
__device__ __forceinline__ void unit3(float& x, float& y, float& z)
{
    float n = sqrtf(dot3_ref(x, y, z, x, y, z));
    x = x / n;
    y = y / n;
    z = z / n;
}

// renderer.rs:59-66 — c = color + sum(lights(p)) in list order, then (x) reflectiveness of element k
__device__ __forceinline__ void shade(const RenderArgs& A, bool active, int k, float px, float py, float pz, float& cr, float& cg, float& cb)
{
    for (int l = 0; l < A.n_lights; ++l) {
        const float4 lp = A.lpos[l];
        const float4 lc = A.lcol[l];
        if (lc.w != 0.0f) {  // light/ambient.rs:23-25
            cr = cr + lc.x;
            cg = cg + lc.y;
            cb = cb + lc.z;
        } else {  // light/point.rs:25-30
            float vx = px - lp.x, vy = py - lp.y, vz = pz - lp.z;
            float dist = sqrtf(dot3_ref(vx, vy, vz, vx, vy, vz));
            float in = lp.w / (dist * dist);
            const bool some = !(in <= 1.1920929e-7f);
            if (some) {
                cr = cr + in * lc.x;
                cg = cg + in * lc.y;
                cb = cb + in * lc.z;
            }
        }
    }
    const float4 rf = A.refl[active ? k : 0];
    cr = cr * rf.x;
    cg = cg * rf.y;
    cb = cb * rf.z;
}

// ---- general-origin rays (bounces) ---------------------------------------------------------------------------------------------
// Conservative filter for a ray (o, u) against sphere (c, r): the reference's own formula (sphere.rs:27-29)
//     q = o - c ;  delta = (u.q)^2 - (|q|^2 - r^2)
// expanded around the scene centre B (A.ball) so that no per-pair subtraction is left:  o' = o - B, c' = c - B,
//     a' = u'.o' - u'.c'                    = fma(-u'x, c'x, fma(-u'y, c'y, fma(-u'z, c'z, A0)))       A0 = u'.o'   (per ray)
//     W  = |o'|^2 - 2 o'.c' + |c'|^2 - rho  = fma(-2o'x, c'x, fma(-2o'y, c'y, fma(-2o'z, c'z, O2 + Kc)))
//     g  = fma(a', a', -W)    evaluated as  h = fma(a', a', -(W - O2)) >= O2         7 FMAs per (ray, sphere), 14 flops
// Margins: u' = u (1 + 2^-16) and rho = r^2 (1+eta)/(1-eta), eta = 2^-17, cover the difference between fused and unfused evaluation of
// the reference's expression (< 48 u (|q|^2 + r^2), u = 2^-24) as before; the expansion adds cancellation error <= 52 u (|o'|^2 + |c'|^2)
// (roundings of o', c', O2, Kc, the sum and the seven FMAs), covered per ray by O2 = |o'|^2 (1 - 64u) and per sphere by
// Kc = |c'|^2 (1 - 64u) - rho, rounded down (scene_prep.cpp).  So g >= 0 whenever the reference's delta >= 0; candidates run the exact test.
constexpr float kGenDirScale = 1.0f + 1.52587890625e-05f;  // 1 + 2^-16
constexpr float kGenO2Scale = 1.0f - 3.814697265625e-06f;  // 1 - 64u
#ifndef RT_GEN_BLOCK
#define RT_GEN_BLOCK 8
#endif
constexpr int kGenBlock = RT_GEN_BLOCK;                    // spheres per wave vote in the general loops

// h(ray, sphere) = a'^2 - T with T = W - O2: the pair is a candidate iff h >= O2 (= g >= 0), so the per-ray term O2 never enters the
// per-pair arithmetic: 7 FMAs per (ray, sphere), and the block vote compares each ray's maximum of h with its own O2.
template <int P>
__device__ __forceinline__ float general_filter(int p, const float (&mx)[P], const float (&my)[P], const float (&mz)[P], const float (&fx)[P],
                                                const float (&fy)[P], const float (&fz)[P], const float (&a0)[P], float cx, float cy, float cz, float kc)
{
    const float a = __builtin_fmaf(fx[p], cx, __builtin_fmaf(fy[p], cy, __builtin_fmaf(fz[p], cz, a0[p])));
    const float t = __builtin_fmaf(mx[p], cx, __builtin_fmaf(my[p], cy, __builtin_fmaf(mz[p], cz, kc)));
    return __builtin_fmaf(a, a, -t);
}

// display.rs:98-100 + renderer.rs:43 — primary ray direction of pixel (x = column j, y = row i): exact reference arithmetic
__device__ __forceinline__ void primary_dir(const RenderArgs& A, uint32_t x, uint32_t y, float& dx, float& dy, float& dz)
{
    const float v0 = 2.0f * ((float)x / (float)A.w) - 1.0f;
    const float v1 = 2.0f * ((float)y / (float)A.h) - 1.0f;
    const float v2 = 2.0f * (1.0f / 1.0f) - 1.0f;
    const float v3 = v2;
    dx = dot4_ref(A.m + 0, v0, v1, v2, v3);
    dy = dot4_ref(A.m + 4, v0, v1, v2, v3);
    dz = dot4_ref(A.m + 8, v0, v1, v2, v3);
    unit3(dx, dy, dz);
}

// The same, also returning |pos| as unit() computes it (vec3.rs:195-197): the culled kernel bounds the rounding error of a direction with it
__device__ __forceinline__ void primary_dir_norm(const RenderArgs& A, uint32_t x, uint32_t y, float& dx, float& dy, float& dz, float& n)
{
    const float v0 = 2.0f * ((float)x / (float)A.w) - 1.0f;
    const float v1 = 2.0f * ((float)y / (float)A.h) - 1.0f;
    const float v2 = 2.0f * (1.0f / 1.0f) - 1.0f;
    const float v3 = v2;
    dx = dot4_ref(A.m + 0, v0, v1, v2, v3);
    dy = dot4_ref(A.m + 4, v0, v1, v2, v3);
    dz = dot4_ref(A.m + 8, v0, v1, v2, v3);
    n = sqrtf(dot3_ref(dx, dy, dz, dx, dy, dz));
    dx = dx / n;
    dy = dy / n;
    dz = dz / n;
}

__device__ __forceinline__ void store_pixel(const RenderArgs& A, size_t pix, float cr, float cg, float cb)
{
    const uint32_t r8 = quantise(cr), g8 = quantise(cg), b8 = quantise(cb);
    if (A.flags & RT_OUT_RGB8) {
        uint8_t* o = A.out + pix * 3;
        o[0] = (uint8_t)r8;
        o[1] = (uint8_t)g8;
        o[2] = (uint8_t)b8;
    } else {
        reinterpret_cast<uint32_t*>(A.out)[pix] = r8 | (g8 << 8) | (b8 << 16) | 0xff000000u;
    }
}

// The end of a tile, given every pixel's closest hit: EMIT = false: shade (depth 1, no shadows) and store the pixel — the whole reference
// path.  EMIT = true: store black, and append every hit as a record (pixel, element, t) to a dense list for the general-ray kernels
// below (bounces / shadow rays then run with full lanes instead of the frame's ~20 % coverage).
// `s_agg` (EMIT only; null = one atomic per wave): kWaves + 1 words of LDS through which the four waves of the workgroup reserve their
// records with ONE atomic (two workgroup barriers: every wave of the workgroup must arrive here) — returning atomics on one address
// complete at only ~40-80 M/s chip-wide, which bounds a kernel whose tiles take microseconds (render_cull_kernel).
template <int P, bool EMIT>
__device__ __forceinline__ void finish_tile(const RenderArgs& A, int lane, uint32_t inb, uint32_t x_base, uint32_t y_base, const float (&dx)[P],
                                            const float (&dy)[P], const float (&dz)[P], const float (&best_t)[P], const int (&best_i)[P],
                                            uint32_t* s_agg = nullptr)
{
    if (!EMIT) {
        // ---- shading and store (depth 1, reference semantics) ----
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const bool hit = best_i[p] >= 0;
            // renderer.rs:57, object/mod.rs:61-63 — origin + t * direction (origin = 0)
            const float px = 0.0f + best_t[p] * dx[p], py = 0.0f + best_t[p] * dy[p], pz = 0.0f + best_t[p] * dz[p];
            float r = 0.0f, g = 0.0f, b = 0.0f;
            if (__builtin_amdgcn_ballot_w64(hit) != 0ull) shade(A, hit, best_i[p], px, py, pz, r, g, b);  // most waves see only background
            if (!((inb >> p) & 1u)) continue;
            const uint32_t y = y_base + kLY * p;
            const size_t pix = (size_t)(y - A.y0) * A.w + x_base;
            store_pixel(A, pix, hit ? r : 0.0f, hit ? g : 0.0f, hit ? b : 0.0f);
            if (A.hit_idx) A.hit_idx[pix] = best_i[p];
        }
    } else {
        // ---- black background + dense hit records: one atomic per wave, order inside the list is irrelevant to the image ----
        uint32_t total = 0;
        uint64_t hitmask[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            hitmask[p] = __builtin_amdgcn_ballot_w64(((inb >> p) & 1u) && best_i[p] >= 0);
            total += (uint32_t)__builtin_popcountll(hitmask[p]);
        }
        uint32_t base = 0;
        if (s_agg) {
            const int wave = threadIdx.x >> 6;
            if (lane == 0) s_agg[wave] = total;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t sum = 0;
                for (int i = 0; i < kWaves; ++i) sum += s_agg[i];
                s_agg[kWaves] = sum ? atomicAdd(A.rec_count, sum) : 0u;
            }
            __syncthreads();
            base = s_agg[kWaves];
            for (int i = 0; i < wave; ++i) base += s_agg[i];
        } else if (lane == 0 && total) {
            base = atomicAdd(A.rec_count, total);
        }
        base = __builtin_amdgcn_readfirstlane(base);
        uint32_t off = base;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const bool in_frame = (inb >> p) & 1u;
            const uint32_t y = y_base + kLY * p;
            const size_t pix = (size_t)(y - A.y0) * A.w + x_base;
            if (in_frame) {
                store_pixel(A, pix, 0.0f, 0.0f, 0.0f);
                if (A.hit_idx) A.hit_idx[pix] = best_i[p];
                if (best_i[p] >= 0) {
                    const uint32_t r = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(hitmask[p] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hitmask[p], 0u));
                    A.rec_pix[r] = (uint32_t)pix;
                    A.rec_idx[r] = best_i[p];
                    A.rec_t[r] = best_t[p];
                }
            }
            off += (uint32_t)__builtin_popcountll(hitmask[p]);
        }
        if (s_agg && A.sort_bins) {
            // the histogram of the first record sort (sort_records_kernel<false>), taken here while the elements are in registers: one
            // atomic (no return value) per distinct element of a ray slot; launch_render zeroes the bins before this kernel and skips
            // the histogram launch of iteration 0
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const bool hit = ((inb >> p) & 1u) && best_i[p] >= 0;
                const int key = hit ? A.rank[best_i[p]] : -1;
                uint64_t todo = hitmask[p];
                while (todo) {
                    const int k0 = __builtin_amdgcn_readlane(key, __builtin_ctzll(todo));
                    const uint64_t same = __builtin_amdgcn_ballot_w64(key == k0);
                    if (lane == 0) atomicAdd(A.sort_bins + k0, (uint32_t)__builtin_popcountll(same));
                    todo &= ~same;
                }
            }
        }
    }
}

// Primary rays of ONE tile (a workgroup's 4 waves side by side: 64 x (4*PY) pixels, PY pixels per lane in one column): closest
// hit over the whole list.  EMIT = false: shade (depth 1, no shadows) and store the pixel — the whole reference path.
// EMIT = true: store black, and append every hit as a record (pixel, element, t) to a dense list for the general-ray kernels below
// (bounces / shadow rays then run with full lanes instead of the frame's ~20 % coverage).
// `staged`: the whole list (one chunk) already sits in LDS; otherwise the list is re-staged chunk by chunk (workgroup barriers).
template <int PY, int B, bool EMIT>
__device__ __forceinline__ void render_tile(const RenderArgs& A, float* s_x, float* s_y, float* s_z, bool staged, uint32_t x_base, uint32_t y_base)
{
    constexpr int P = PY;
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    float dx[P], dy[P], dz[P], best_t[P];
    int best_i[P];
    uint32_t inb = 0;

    // ---- primary rays ----
#pragma unroll
    for (int p = 0; p < P; ++p) {
        uint32_t x = x_base;
        uint32_t y = y_base + kLY * p;
        if (x < A.w && y < A.y1) inb |= 1u << p;
        x = min(x, A.w - 1u);
        y = min(y, A.y1 - 1u);
        primary_dir(A, x, y, dx[p], dy[p], dz[p]);
        best_t[p] = 0.0f;
        best_i[p] = -1;
    }

    // ---- spheres the filter cannot cover: exact test for every pixel ----
    for (int a = 0; a < A.n_always; ++a) {
        const int k = A.always[a];
        const float4 s = A.sph[k];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float t;
            bool h = hit_primary_exact(dx[p], dy[p], dz[p], s, t);
            take_closest(h, t, k, best_t[p], best_i[p]);
        }
    }

    // ---- planes (extension): no filter, exact test for every pixel ----
    for (int a = 0; a < A.n_planes; ++a) {
        const int k = A.planes[a];
        const float4 p0 = A.sph_r[k], n = A.pln[k];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float t;
            bool h = hit_plane_exact(0.0f, 0.0f, 0.0f, dx[p], dy[p], dz[p], p0, n, t);
            take_closest(h, t, k, best_t[p], best_i[p]);
        }
    }

    // ---- hot loop: conservative 3-FMA filter over the whole list, exact test only for candidate blocks ----
    for (int c0 = 0; c0 < A.n_pad; c0 += kChunk) {
        const int cn = min(kChunk, A.n_pad - c0);
        if (!staged) {
            __syncthreads();  // every wave is done reading the previous chunk
            for (int i = tid; i < cn; i += kWaves * 64) {
                const float4 q = A.filt[c0 + i];
                s_x[i] = q.x;
                s_y[i] = q.y;
                s_z[i] = q.z;
            }
            __syncthreads();
        }
        for (int s0 = 0; s0 < cn; s0 += B) {
            float m = -1.0f;
            float qx[B], qy[B], qz[B];
#pragma unroll
            for (int g = 0; g < B; g += 4) {  // wave-uniform address: broadcast ds_read_b128, four spheres per read
                const float4 vx = *reinterpret_cast<const float4*>(&s_x[s0 + g]);
                const float4 vy = *reinterpret_cast<const float4*>(&s_y[s0 + g]);
                const float4 vz = *reinterpret_cast<const float4*>(&s_z[s0 + g]);
                qx[g] = vx.x; qx[g + 1] = vx.y; qx[g + 2] = vx.z; qx[g + 3] = vx.w;
                qy[g] = vy.x; qy[g + 1] = vy.y; qy[g + 2] = vy.z; qy[g + 3] = vy.w;
                qz[g] = vz.x; qz[g + 1] = vz.y; qz[g + 2] = vz.z; qz[g + 3] = vz.w;
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    // 3 x v_fma_f32 per (pixel, sphere); the block maximum folds two results per v_max3_f32
                    float f = __builtin_fmaf(dx[p], qx[b], __builtin_fmaf(dy[p], qy[b], __builtin_fmaf(dz[p], qz[b], -1.0f)));
                    m = __builtin_fmaxf(m, f);
                }
            }
            if (__builtin_amdgcn_ballot_w64(m >= 0.0f) != 0ull) {
                // rare: some lane has a candidate in this block. Find the spheres (constants still in registers), run the
                // reference's exact test.
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    float mb = -1.0f;
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        mb = __builtin_fmaxf(mb, __builtin_fmaf(dx[p], qx[b], __builtin_fmaf(dy[p], qy[b], __builtin_fmaf(dz[p], qz[b], -1.0f))));
                    if (__builtin_amdgcn_ballot_w64(mb >= 0.0f) == 0ull) continue;
                    const int k = c0 + s0 + b;
                    const float4 s = A.sph[k];
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        float t;
                        bool h = hit_primary_exact(dx[p], dy[p], dz[p], s, t);
                        take_closest(h, t, k, best_t[p], best_i[p]);
                    }
                }
            }
        }
    }

    finish_tile<P, EMIT>(A, lane, inb, x_base, y_base, dx, dy, dz, best_t, best_i);
}

// The primary-ray kernel: a PERSISTENT grid (one workgroup per resident slot of the chip, launch_render) whose workgroups pull
// tiles from a queue (A.tile_ctr[0], one atomic per tile).  The band is cut into three sections of decreasing tile height —
// 16-row tiles (4 pixels per lane, the efficient shape) first, then 8-row, then 4-row tiles — so the work units shrink towards
// the end of the launch and every CU runs dry at nearly the same time; a static grid of equal tiles instead leaves half the chip
// idle during its last partial round (profiles/r01_tuning.md).  The last workgroup to leave re-arms the queue for the next
// launch that uses this counter slot (no host memset; also safe to replay).
template <int B, bool EMIT>
__global__ __launch_bounds__(kWaves * 64) void render_kernel(const RenderArgs A)
{
    // filter constants of the current chunk, SoA so that four spheres' x (y, z) come back in one broadcast ds_read_b128
    __shared__ __attribute__((aligned(16))) float s_x[kChunk], s_y[kChunk], s_z[kChunk];
    __shared__ uint32_t s_tile;

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const uint32_t lx = (uint32_t)(wave * kLX + (lane & (kLX - 1)));  // column inside the 64-pixel strip
    const uint32_t ly = (uint32_t)(lane >> 4);

    const bool staged = A.n_pad <= kChunk;  // the whole list fits one chunk: stage it once per workgroup, not once per tile
    if (staged) {
        for (int i = tid; i < A.n_pad; i += kWaves * 64) {
            const float4 q = A.filt[i];
            s_x[i] = q.x;
            s_y[i] = q.y;
            s_z[i] = q.z;
        }
    }
    const uint32_t n_tiles = A.tiles4 + A.tiles2 + A.tiles1;
    for (uint32_t round = 0;; ++round) {
        __syncthreads();  // s_tile of the previous round has been read by everyone (first round: the staged list is visible)
        if (A.static_grid) {  // small band: one tile per workgroup, no queue (launch_primary)
            if (tid == 0) s_tile = round == 0 ? blockIdx.x : n_tiles;
        } else if (tid == 0) {
            s_tile = atomicAdd(A.tile_ctr, 1u);
        }
        __syncthreads();
        uint32_t t = s_tile;
        if (t >= n_tiles) break;
        if (t < A.tiles4) {
            render_tile<4, B, EMIT>(A, s_x, s_y, s_z, staged, (t % A.strips) * (kWaves * kLX) + lx, A.y0 + (t / A.strips) * (4 * kLY) + ly);
        } else if ((t -= A.tiles4) < A.tiles2) {
            render_tile<2, B, EMIT>(A, s_x, s_y, s_z, staged, (t % A.strips) * (kWaves * kLX) + lx, A.y0 + A.rows4 + (t / A.strips) * (2 * kLY) + ly);
        } else {
            t -= A.tiles2;
            render_tile<1, B, EMIT>(A, s_x, s_y, s_z, staged, (t % A.strips) * (kWaves * kLX) + lx, A.y0 + A.rows4 + A.rows2 + (t / A.strips) * kLY + ly);
        }
    }
    // every workgroup does exactly one failing fetch and then reports here, so the last one to arrive knows that nobody will
    // touch the queue again and resets it; the end of the kernel publishes the two stores to the next launch
    if (tid == 0 && !A.static_grid) {
        if (atomicAdd(A.tile_ctr + 1, 1u) == gridDim.x - 1u) {
            atomicExch(A.tile_ctr, 0u);
            atomicExch(A.tile_ctr + 1, 0u);
        }
    }
}

// ---- general rays (depth > 1, RT_SHADOWS): one pass of two kernels per iteration of renderer.rs:46 over the dense hit records ----
// Ray state of record r at iteration A.iter: iteration 0 = the primary ray (origin 0, direction recomputed from the pixel with
// the reference's arithmetic); later iterations = the bounce ray that shade_iter_kernel stored (planes of A.rec_state).
enum { kStOx = 0, kStOy, kStOz, kStDx, kStDy, kStDz, kStCr, kStCg, kStCb, kStPlanes };

__device__ __forceinline__ void load_ray(const RenderArgs& A, uint32_t r, bool ok, uint32_t pix, float& ox, float& oy, float& oz, float& dx, float& dy, float& dz)
{
    if (A.iter == 0u) {
        const uint32_t y = A.y0 + pix / A.w, x = pix % A.w;
        primary_dir(A, x, y, dx, dy, dz);
        ox = oy = oz = 0.0f;
    } else {
        const size_t c = A.rec_cap, i = ok ? r : 0u;
        ox = A.rec_state[kStOx * c + i]; oy = A.rec_state[kStOy * c + i]; oz = A.rec_state[kStOz * c + i];
        dx = A.rec_state[kStDx * c + i]; dy = A.rec_state[kStDy * c + i]; dz = A.rec_state[kStDz * c + i];
    }
}

// ---- RT_SHADOWS (EXTENSION — the reference has no occlusion, light/point.rs:25-30) -------------------------------------------------
// Spatial sort of the live hit records (a counting sort on the spatial rank of the element each record sits on, scene_prep.cpp:
// spatial_order): A.rec_perm lists the live records so that consecutive entries lie on the same or on neighbouring spheres; their
// shadow rays toward one light then form a narrow cone, which is what the any-hit hierarchy of shadow_mask_kernel feeds on.  Also
// drops dead records (rec_idx < 0).  Three small kernels per iteration of renderer.rs:46: histogram, exclusive scan, scatter.
__device__ __forceinline__ int sort_key(const RenderArgs& A, uint32_t r, uint32_t count)
{
    if (r >= count) return -1;
    const int idx = A.rec_idx[r];
    return idx >= 0 ? A.rank[idx] : -1;
}

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)  // set bits of `mask` in lanes lower than this one
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// SCATTER = false: A.sort_bins[key] += records with that key (bins zeroed before).  SCATTER = true: bins hold the exclusive prefix;
// every record takes the next free position of its bin.  One atomic per distinct key of a wave (neighbouring records share elements).
template <bool SCATTER>
__global__ __launch_bounds__(256) void sort_records_kernel(const RenderArgs A)
{
    const uint32_t count = *A.rec_count;
    const int lane = threadIdx.x & 63;
    for (uint32_t base = (blockIdx.x * 256u + (threadIdx.x & ~63u)); base < count; base += gridDim.x * 256u) {
        const uint32_t r = base + (uint32_t)lane;
        const int key = sort_key(A, r, count);
        uint32_t dst = 0;
        uint64_t todo = __builtin_amdgcn_ballot_w64(key >= 0);
        while (todo) {
            const int lead = __builtin_ctzll(todo);
            const int k0 = __builtin_amdgcn_readlane(key, lead);
            const uint64_t same = __builtin_amdgcn_ballot_w64(key == k0);
            uint32_t at = 0;
            if (lane == lead) at = atomicAdd(A.sort_bins + k0, (uint32_t)__builtin_popcountll(same));
            if (SCATTER) {
                at = (uint32_t)__builtin_amdgcn_readlane((int)at, lead);
                if (key == k0) dst = at + lanes_below(same);
            }
            todo &= ~same;
        }
        if (SCATTER) {
            // the hit point of the record (renderer.rs:57), once per iteration instead of once per light, stored in SORTED order so that
            // shadow_mask_kernel reads it with unit stride
            const bool ok = key >= 0;
            float ox, oy, oz, dx, dy, dz;
            load_ray(A, r, ok, ok ? A.rec_pix[r] : 0u, ox, oy, oz, dx, dy, dz);
            const float tt = ok ? A.rec_t[r] : 0.0f;
            if (ok) {
                const size_t c = A.rec_cap;
                A.rec_perm[dst] = r;
                A.rec_hit[dst] = ox + tt * dx;
                A.rec_hit[c + dst] = oy + tt * dy;
                A.rec_hit[2 * c + dst] = oz + tt * dz;
            }
        }
    }
}

// Exclusive scan of A.sort_bins[0, n) in place (one workgroup); the total = number of live records goes to A.rec_count[1].
__global__ __launch_bounds__(256) void sort_scan_kernel(const RenderArgs A)
{
    __shared__ uint32_t s_part[256];
    const uint32_t n = (uint32_t)A.n, tid = threadIdx.x;
    const uint32_t per = (n + 255u) / 256u;
    const uint32_t lo = min(n, tid * per), hi = min(n, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += A.sort_bins[i];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 256; ++i) {
            const uint32_t v = s_part[i];
            s_part[i] = run;
            run += v;
        }
        A.rec_count[1] = run;
    }
    __syncthreads();
    uint32_t run = s_part[tid];
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t v = A.sort_bins[i];
        A.sort_bins[i] = run;
        run += v;
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m; m >>= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)  // NaN operands are ignored (fmaxf)
{
#pragma unroll
    for (int m = 32; m; m >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int m = 32; m; m >>= 1) v = __builtin_fminf(v, __shfl_xor(v, m, 64));
    return v;
}

constexpr float kConeP1 = 1.0f - 3.814697265625e-06f;  // P1 = 1 - 64u of the member test (scene_prep.cpp: prepare_light_filters)

// The cone of the directions a wave holds: `fin` = the rays of `mask` whose direction (dx, dy, dz)[p] is finite (the others miss every
// exact test and need no candidates); unit axis a = normalised sum of those directions; sinT >= sin and cosT <= cos of the largest
// angle between a and one of them (|a x d| is accurate for small angles; 1e-6 relative and absolute slack).  Returns true if they
// all lie within 60 degrees of a; false (NaN axis included) means "no cone": the caller keeps every sphere.
template <int P>
__device__ __forceinline__ bool wave_cone(uint32_t mask, const float (&dx)[P], const float (&dy)[P], const float (&dz)[P], uint32_t& fin, float& ax,
                                          float& ay, float& az, float& sinT, float& cosT)
{
    ax = ay = az = 0.0f;
    fin = 0u;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (((mask >> p) & 1u) && __builtin_isfinite(dx[p]) && __builtin_isfinite(dy[p]) && __builtin_isfinite(dz[p])) {
            fin |= 1u << p;
            ax = ax + dx[p];
            ay = ay + dy[p];
            az = az + dz[p];
        }
    }
    ax = wave_sum(ax);
    ay = wave_sum(ay);
    az = wave_sum(az);
    const float an = sqrtf(ax * ax + ay * ay + az * az);
    ax = ax / an;
    ay = ay / an;
    az = az / an;
    float cmin = INFINITY, smax = 0.0f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (!((fin >> p) & 1u)) continue;
        const float c = ax * dx[p] + ay * dy[p] + az * dz[p];
        const float kx = ay * dz[p] - az * dy[p], ky = az * dx[p] - ax * dz[p], kz = ax * dy[p] - ay * dx[p];
        cmin = __builtin_fminf(cmin, c);
        smax = __builtin_fmaxf(smax, sqrtf(kx * kx + ky * ky + kz * kz));
    }
    cmin = wave_min(cmin);
    smax = wave_max(smax);
    sinT = smax * (1.0f + 1.0e-6f) + 1.0e-6f;
    cosT = sqrtf(__builtin_fmaxf(0.0f, 1.0f - sinT * sinT)) * (1.0f - 1.0e-6f);
    return an > 0.0f && an < INFINITY && cmin > 0.5f && cmin <= 2.0f;
}

// The cone of a TILE of primary rays without any cross-lane reduction: pos = M v is affine in the pixel coordinates (display.rs:98-100), so
// in exact arithmetic the directions of a wave's pixel rectangle lie in the convex cone spanned by its four CORNER directions, and for
// every x of that cone cos(angle(a, x)) >= the smallest cos(angle(a, corner)) as long as all four are positive.  The corners are held
// by lanes 0 and 15 (first row, ray slot 0) and 48 and 63 (last row, ray slot P - 1) — clamped pixels at the frame's edge included —
// and are read with v_readlane; everything after that is wave-uniform.
// Rounding: the f32 evaluation of display.rs:98-100 puts every pos within  e = 6u (S0^2 + S1^2 + S2^2)^(1/2),  S_r = sum |m_r.|, of the
// affine model (v: 3u absolute; the two products 4u |m_r0|, 4u |m_r1|; the three sums u each of at most S_r; u = 2^-24) — that is NOT small
// against |pos| when the rows of a caller's matrix cancel — and a point of the parallelogram is at least half as far from the origin
// as the nearest corner (its corners lie within 60 degrees of the axis), so a computed direction is within  delta = 2 e / min |pos_corner|
// of the model's, a corner's own included.  `err` = 8u |S| comes in from the kernel, the corner norms from primary_dir_norm;
// sinT carries 3 x 2.5 err / min|pos| + 4e-6 (normalisation, cross product) of slack.  Returns false ("no cone": the caller keeps every
// sphere) if a corner is not finite, the corners do not fit within 60 degrees of their normalised sum, or delta is not small.
template <int P>
__device__ __forceinline__ bool tile_cone(const float (&dx)[P], const float (&dy)[P], const float (&dz)[P], const float (&nrm)[P], float err,
                                          float& ax, float& ay, float& az, float& sinT, float& cosT)
{
    auto corner = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    const float gx[4] = {corner(dx[0], 0), corner(dx[0], 15), corner(dx[P - 1], 48), corner(dx[P - 1], 63)};
    const float gy[4] = {corner(dy[0], 0), corner(dy[0], 15), corner(dy[P - 1], 48), corner(dy[P - 1], 63)};
    const float gz[4] = {corner(dz[0], 0), corner(dz[0], 15), corner(dz[P - 1], 48), corner(dz[P - 1], 63)};
    const float nmin = __builtin_fminf(__builtin_fminf(corner(nrm[0], 0), corner(nrm[0], 15)), __builtin_fminf(corner(nrm[P - 1], 48), corner(nrm[P - 1], 63)));
    const float delta = 2.5f * err / nmin;
    ax = (gx[0] + gx[1]) + (gx[2] + gx[3]);
    ay = (gy[0] + gy[1]) + (gy[2] + gy[3]);
    az = (gz[0] + gz[1]) + (gz[2] + gz[3]);
    const float an = sqrtf(ax * ax + ay * ay + az * az);
    ax = ax / an;
    ay = ay / an;
    az = az / an;
    float cmin = INFINITY, smax = 0.0f;
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        finite = finite && __builtin_isfinite(gx[i]) && __builtin_isfinite(gy[i]) && __builtin_isfinite(gz[i]);
        const float c = ax * gx[i] + ay * gy[i] + az * gz[i];
        const float kx = ay * gz[i] - az * gy[i], ky = az * gx[i] - ax * gz[i], kz = ax * gy[i] - ay * gx[i];
        cmin = __builtin_fminf(cmin, c);
        smax = __builtin_fmaxf(smax, sqrtf(kx * kx + ky * ky + kz * kz));
    }
    sinT = smax * (1.0f + 1.0e-6f) + (4.0e-6f + 3.0f * delta);
    cosT = sqrtf(__builtin_fmaxf(0.0f, 1.0f - sinT * sinT)) * (1.0f - 1.0e-6f);
    return finite && an > 0.0f && an < INFINITY && cmin > 0.5f && cmin <= 2.0f && delta < 0.05f;
}

// ---- RT_CULL: the primary rays through the cone hierarchy ------------------------------------------------------------------------------
// Same image as render_kernel, byte for byte, without the walk over the whole list: a wave's tile (16 x 16 pixels, 4 per lane) is a
// narrow cone of directions from the common origin 0, so the votes of the shadow rays apply with the "light" at the origin and the
// primary filter's own constants (scene_prep.cpp: prepare_primary_cull; spheres ordered by the direction they are seen in, clusters of
// 16).  Cluster vote -> member vote -> a per-wave list in LDS (c', element index) -> the 3-FMA filter of the hot loop, 8 spheres per
// wave vote -> the reference's exact test for candidates -> closest hit as the lexicographic minimum of (t, index), so the list order
// is immaterial.  Neither vote drops a sphere the per-pixel filter would pass.  Spheres on the always-list and planes are tested
// exactly for every pixel, as in render_tile.  A plain grid, one tile (4 waves side by side: 64 x 16 pixels) per workgroup: tiles are
// microseconds long, a queue would be bound by its atomic.  The waves are independent (no barrier) until they reserve their hit records.
#ifndef RT_CULL_P
#define RT_CULL_P 4
#endif
constexpr int kCullP = RT_CULL_P;  // pixels per lane (one column, kLY rows apart) of render_cull_kernel: a wave's tile is 16 x (4 * kCullP) pixels
template <bool EMIT, int CAP>
__global__ __launch_bounds__(kWaves * 64) void render_cull_kernel(const RenderArgs A)
{
    constexpr int P = kCullP, B = 8;
    static_assert(CAP % B == 0 && CAP >= 128, "list geometry");
    __shared__ float4 s_list[kWaves][CAP];  // per wave: (c'x, c'y, c'z, element index as bits)
    __shared__ int s_open[kWaves][256];     // per wave: the clusters the current vote opened
    __shared__ uint32_t s_agg[kWaves + 1];  // EMIT: the workgroup's record reservation (finish_tile)
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    float4* const s_l = s_list[wave];
    int* const s_cl = s_open[wave];
    const uint32_t t = blockIdx.x;
    const uint32_t x_base = (t % A.strips) * (kWaves * kLX) + (uint32_t)(wave * kLX + (lane & (kLX - 1)));
    const uint32_t y_base = A.y0 + (t / A.strips) * (P * kLY) + (uint32_t)(lane >> 4);
    // (wave-uniform) false: this wave's 16 columns lie beyond the frame; it only keeps the workgroup's barriers company
    const bool in_frame = (t % A.strips) * (kWaves * kLX) + (uint32_t)(wave * kLX) < A.w;

    float dx[P], dy[P], dz[P], nrm[P], best_t[P];
    int best_i[P];
    uint32_t inb = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        uint32_t x = x_base;
        uint32_t y = y_base + kLY * p;
        if (x < A.w && y < A.y1) inb |= 1u << p;
        x = min(x, A.w - 1u);
        y = min(y, A.y1 - 1u);
        primary_dir_norm(A, x, y, dx[p], dy[p], dz[p], nrm[p]);
        best_t[p] = 0.0f;
        best_i[p] = -1;
    }
    for (int a = 0; a < A.n_always; ++a) {  // spheres the filter cannot cover: exact test for every pixel
        const int k = A.always[a];
        const float4 s = A.sph[k];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float tt;
            const bool h = hit_primary_exact(dx[p], dy[p], dz[p], s, tt);
            take_closest(h, tt, k, best_t[p], best_i[p]);
        }
    }
    for (int a = 0; a < A.n_planes; ++a) {  // planes (extension): no filter, exact test for every pixel
        const int k = A.planes[a];
        const float4 p0 = A.sph_r[k], n = A.pln[k];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float tt;
            const bool h = hit_plane_exact(0.0f, 0.0f, 0.0f, dx[p], dy[p], dz[p], p0, n, tt);
            take_closest(h, tt, k, best_t[p], best_i[p]);
        }
    }
    // ---- the cone of the tile's directions (a ray with a non-finite direction misses every exact test and needs no candidates) ----
    float ax, ay, az, sinT, cosT;
    float err;  // 8u |S|: bound (with margin) of the rounding error of a computed pos, see tile_cone
    {
        const float s0 = (__builtin_fabsf(A.m[0]) + __builtin_fabsf(A.m[1])) + (__builtin_fabsf(A.m[2]) + __builtin_fabsf(A.m[3]));
        const float s1 = (__builtin_fabsf(A.m[4]) + __builtin_fabsf(A.m[5])) + (__builtin_fabsf(A.m[6]) + __builtin_fabsf(A.m[7]));
        const float s2 = (__builtin_fabsf(A.m[8]) + __builtin_fabsf(A.m[9])) + (__builtin_fabsf(A.m[10]) + __builtin_fabsf(A.m[11]));
        err = 4.76837158203125e-07f * sqrtf(s0 * s0 + s1 * s1 + s2 * s2);
    }
    const bool wide = !tile_cone<P>(dx, dy, dz, nrm, err, ax, ay, az, sinT, cosT);
    if (in_frame) {
        int count = 0;
        auto walk = [&]() {
            const int padded = (count + B - 1) / B * B;
            if (count + lane < padded) s_l[count + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // entries that never pass
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int s0 = 0; s0 < padded; s0 += B) {
                float m = -1.0f;
                float4 q[B];
#pragma unroll
                for (int b = 0; b < B; ++b) q[b] = s_l[s0 + b];  // wave-uniform address: broadcast ds_read_b128
#pragma unroll
                for (int b = 0; b < B; ++b)
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        m = __builtin_fmaxf(m, __builtin_fmaf(dx[p], q[b].x, __builtin_fmaf(dy[p], q[b].y, __builtin_fmaf(dz[p], q[b].z, -1.0f))));
                if (__builtin_amdgcn_ballot_w64(m >= 0.0f) == 0ull) continue;
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    float mb = -1.0f;
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        mb = __builtin_fmaxf(mb, __builtin_fmaf(dx[p], q[b].x, __builtin_fmaf(dy[p], q[b].y, __builtin_fmaf(dz[p], q[b].z, -1.0f))));
                    if (__builtin_amdgcn_ballot_w64(mb >= 0.0f) == 0ull) continue;
                    const int k = __float_as_int(q[b].w);
                    const float4 s = A.sph[k];
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        float tt;
                        const bool h = hit_primary_exact(dx[p], dy[p], dz[p], s, tt);
                        take_closest(h, tt, k, best_t[p], best_i[p]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            count = 0;
        };
        const int n_cl = A.n_pad / 16;
        for (int cb = 0; cb < n_cl; cb += 256) {
            // ---- cluster vote: up to four clusters per lane, their loads in flight together; open clusters -> s_cl ----
            float4 ca[4], cs[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = cb + 64 * i + lane;
                ca[i] = c < n_cl ? A.pclA[c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                cs[i] = c < n_cl ? A.pclB[c] : make_float4(2.0f, 0.0f, 0.0f, 0.0f);
            }
            int n_open = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = cb + 64 * i + lane;
                const float dot = __builtin_fmaf(ax, ca[i].x, __builtin_fmaf(ay, ca[i].y, az * ca[i].z));
                const float rhs = __builtin_fmaf(cosT, cs[i].x, -(sinT * cs[i].y));
                const bool pass = c < n_cl && (wide || dot >= rhs - 4.0e-6f);
                const uint64_t mask = __builtin_amdgcn_ballot_w64(pass);
                if (pass) s_cl[n_open + (int)lanes_below(mask)] = c;
                n_open += __builtin_popcountll(mask);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- member vote: four open clusters at a time, 16 lanes each ----
            for (int base = 0; base < n_open; base += 4) {
                const int ci = base + (lane >> 4);
                const int j = (ci < n_open ? s_cl[ci] : 0) * 16 + (lane & 15);
                const float4 q = ci < n_open ? A.pfilt_o[j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                const int k = A.porder[j];
                const float dot = __builtin_fmaf(ax, q.x, __builtin_fmaf(ay, q.y, az * q.z));
                const float rhs = __builtin_fmaf(cosT, kConeP1, -(sinT * q.w));
                const bool entry = q.x != 0.0f || q.y != 0.0f || q.z != 0.0f;  // all zero: no entry at this position
                const bool keep = entry && (wide || dot >= rhs - 8.0e-6f);
                const uint64_t km = __builtin_amdgcn_ballot_w64(keep);
                if (km == 0ull) continue;
                if (keep) s_l[count + (int)lanes_below(km)] = make_float4(q.x, q.y, q.z, __int_as_float(k));
                count += __builtin_popcountll(km);
                if (count > CAP - 64) walk();
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (count > 0) walk();
    }
    finish_tile<P, EMIT>(A, lane, inb, x_base, y_base, dx, dy, dz, best_t, best_i, EMIT ? s_agg : nullptr);
}

// Occlusion bits of every (live record, point light) pair of this iteration, semantics of the oracle's occluded().  A persistent grid
// whose WAVES pull tiles (light l, 64 * P consecutive entries of the spatially sorted record list) from the queue A.tile_ctr, light-major.
// (Several lights per tile, the hit points loaded once, measured slower: one light's constants at a time stay in every CU's caches.)
// All rays toward one light end at the same point: seen from the light they share an origin and the primary loop's 3-FMA filter
// applies with the per-light constants A.lfilt (P rays per lane, broadcast LDS reads, one wave vote per B spheres); candidates run the
// reference's exact general-origin test on the ORIGINAL ray (h, u), so the result is bit-identical to the oracle's.
// Any-hit hierarchy (scene_prep.cpp: prepare_light_filters): the wave bounds its rays' reversed directions by a cone (axis a,
// half-angle theta) and their lengths by sdmax; a cluster of 16 spatially adjacent spheres is opened only if its bounding cone meets
// the wave's and it is not farther from the light than every ray is long; its members are kept only if their own cone meets the
// wave's.  What is kept goes to a per-wave list in LDS and is walked with the 3-FMA filter; the wave leaves a light as soon as all
// its rays are blocked (the early-out).  Neither vote ever drops a sphere that the per-ray filter would pass, so the bits equal those
// of a walk over the whole list.  Waves whose rays do not fit a 60-degree cone (light next to the surface, records on far-apart
// elements) open every cluster.  Result: bit (l & 31) of A.rec_occ[(l >> 5) * rec_cap + r] (atomicOr; the words are zeroed by launch_render).
// `far` marks rays whose hit point lies outside the ball the filter margins were derived for (only possible on planes): a wave that
// holds one tests every sphere exactly.  No workgroup barrier anywhere: the four waves of a workgroup are independent.
template <int P, int B, int CAP, int TB>
__global__ __launch_bounds__(kWaves * 64) void shadow_mask_kernel(const RenderArgs A)
{
    static_assert(CAP % B == 0 && CAP >= 128 && B % 4 == 0, "shortlist geometry");
    __shared__ __attribute__((aligned(16))) float s_list[kWaves][7 * CAP];  // per wave: filter x | y | z, then the sphere (cx, cy, cz, r) per entry
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    float* const s_x = s_list[wave];
    float* const s_y = s_x + CAP;
    float* const s_z = s_x + 2 * CAP;
    float4* const s_s = reinterpret_cast<float4*>(s_x + 3 * CAP);
    __shared__ int s_open[kWaves][256];  // per wave: the clusters the current vote opened
    int* const s_cl = s_open[wave];
    const uint32_t live = A.rec_count[1];
    const uint32_t per_tile = 64 * P;
    const uint32_t groups = (live + per_tile - 1) / per_tile;
    const uint32_t n_tiles = groups * (uint32_t)A.n_lights;
    const int n_cl = A.n_pad / 16;
    // Tiles come from the queue TB at a time: returning atomics on ONE address complete at only ~40-80 M/s chip-wide, a rate the small
    // tiles of this kernel would otherwise be bound by (tiles of either size took 12-14 ns each).  (Giving every wave its first TB
    // tiles statically, to spare the burst of fetches at the start of the launch, measured 6 % slower: profiles/r03_tuning.md.)
    uint32_t t_next = 0, t_stop = 0;
    for (;;) {
        if (t_next == t_stop) {
            uint32_t t0 = 0;
            if (lane == 0) t0 = atomicAdd(A.tile_ctr, (uint32_t)TB);
            t_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)t0);
            t_stop = t_next + TB;
            if (t_next >= n_tiles) break;
        }
        const uint32_t t = t_next++;
        if (t >= n_tiles) continue;
        const int l = (int)(t / groups);  // light-major: waves that run together work on the same light (its constants stay in the caches)
        const uint32_t first = (t % groups) * per_tile;
        const float4 lp = A.lpos[l];
        if (A.lcol[l].w != 0.0f) continue;  // ambient lights are never shadowed
        float hx[P], hy[P], hz[P];
        uint32_t rec[P];
        uint32_t okm = 0u, far = 0u;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t e = first + p * 64 + lane;
            const bool ok = e < live;  // (the sorted list holds live records only)
            const size_t i = ok ? e : 0u;
            rec[p] = A.rec_perm[i];
            if (ok) okm |= 1u << p;
            hx[p] = A.rec_hit[i];  // renderer.rs:57, computed by the scatter kernel
            hy[p] = A.rec_hit[(size_t)A.rec_cap + i];
            hz[p] = A.rec_hit[2 * (size_t)A.rec_cap + i];
            {  // inside the ball the per-light filter margins assume?  (NaN compares false => far)
                const float bx = hx[p] - A.ball[0], by = hy[p] - A.ball[1], bz = hz[p] - A.ball[2];
                if (!(bx * bx + by * by + bz * bz <= A.ball[3])) far |= 1u << p;
            }
        }
        if (!A.n_planes) far = 0u;
        {
            float ux[P], uy[P], uz[P], sd[P], wx[P], wy[P], wz[P];
            uint32_t need = 0u;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                {  // light/point.rs:25-30: a light whose contribution is cut off casts no shadow ray
                    const float vx = hx[p] - lp.x, vy = hy[p] - lp.y, vz = hz[p] - lp.z;
                    const float dist = sqrtf(dot3_ref(vx, vy, vz, vx, vy, vz));
                    const float in = lp.w / (dist * dist);
                    if (((okm >> p) & 1u) && !(in <= 1.1920929e-7f)) need |= 1u << p;
                }
                const float vx = lp.x - hx[p], vy = lp.y - hy[p], vz = lp.z - hz[p];
                sd[p] = sqrtf(dot3_ref(vx, vy, vz, vx, vy, vz));
                ux[p] = vx / sd[p];
                uy[p] = vy / sd[p];
                uz[p] = vz / sd[p];
                const bool on = ((need & ~far) >> p) & 1u;  // dead (and far) rays: w = 0 => f = -1, never a candidate
                wx[p] = on ? -ux[p] : 0.0f;
                wy[p] = on ? -uy[p] : 0.0f;
                wz[p] = on ? -uz[p] : 0.0f;
            }
            uint32_t occ = 0u;
            // the reference's test of sphere s for the ray slots named in `slots` (wave-uniform: a slot none of whose 64 rays is a candidate is skipped)
            auto exact = [&](const float4 s, uint32_t slots = ~0u) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (!((slots >> p) & 1u)) continue;
                    float tt;
                    const bool h = hit_general_exact(hx[p], hy[p], hz[p], ux[p], uy[p], uz[p], s, tt);
                    if (h && tt > RT_SHADOW_EPS && tt < sd[p] && ((need >> p) & 1u)) {
                        occ |= 1u << p;
                        wx[p] = wy[p] = wz[p] = 0.0f;  // decided: stop raising candidates
                    }
                }
            };
            for (int a = 0; a < A.n_planes; ++a) {  // planes: exact test for every ray
                const int k = A.planes[a];
                const float4 p0 = A.sph_r[k], n = A.pln[k];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    float tt;
                    const bool h = hit_plane_exact(hx[p], hy[p], hz[p], ux[p], uy[p], uz[p], p0, n, tt);
                    if (h && tt > RT_SHADOW_EPS && tt < sd[p] && ((need >> p) & 1u)) {
                        occ |= 1u << p;
                        wx[p] = wy[p] = wz[p] = 0.0f;
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64((need & far) != 0u) != 0ull) {  // rare: hit points beyond the filter's ball, every sphere exactly
                for (int k = 0; k < A.n; ++k) exact(A.sph_r[k]);            // (planes and padding entries carry a NaN radius: they miss)
            } else {
                for (int a = A.lalways_off[l]; a < A.lalways_off[l + 1]; ++a) exact(A.sph_r[A.lalways[a]]);
                uint32_t open = need & ~occ;
                if (__builtin_amdgcn_ballot_w64(open != 0u) != 0ull) {
                    // ---- the wave's cone over the reversed directions of its open rays, and the longest of those rays ----
                    float ax, ay, az, sinT, cosT, sdmax = 0.0f;
                    uint32_t fin;
                    const bool wide = !wave_cone<P>(open, wx, wy, wz, fin, ax, ay, az, sinT, cosT);
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        if ((open >> p) & 1u) sdmax = __builtin_fmaxf(sdmax, sd[p]);
                    sdmax = wave_max(sdmax);
#ifdef RT_MASK_STATS
                    if (lane == 0) {
                        atomicAdd(A.tile_ctr + 2, 1u);
                        if (wide) atomicAdd(A.tile_ctr + 3, 1u);
                        atomicAdd(A.tile_ctr + 8, (uint32_t)(sinT * 1000.0f));
                    }
#endif
                    const float4* const clA = A.lclA + (size_t)l * n_cl;
                    const float4* const clB = A.lclB + (size_t)l * n_cl;
                    const float4* const lf = A.lfilt + (size_t)l * A.n_pad;
                    int count = 0;
                    // walk the per-wave list: the 3-FMA filter of the primary loop, B spheres per wave vote; false = every ray is blocked
                    auto walk = [&]() -> bool {
                        const int padded = (count + B - 1) / B * B;
                        if (count + lane < padded) {  // pad the last block with entries that never pass
                            s_x[count + lane] = 0.0f;
                            s_y[count + lane] = 0.0f;
                            s_z[count + lane] = 0.0f;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        for (int s0 = 0; s0 < padded; s0 += B) {
                            float m = -1.0f;
                            float qx[B], qy[B], qz[B];
#pragma unroll
                            for (int g = 0; g < B; g += 4) {  // wave-uniform addresses: broadcast ds_read_b128, four spheres per read
                                const float4 vx = *reinterpret_cast<const float4*>(&s_x[s0 + g]);
                                const float4 vy = *reinterpret_cast<const float4*>(&s_y[s0 + g]);
                                const float4 vz = *reinterpret_cast<const float4*>(&s_z[s0 + g]);
                                qx[g] = vx.x; qx[g + 1] = vx.y; qx[g + 2] = vx.z; qx[g + 3] = vx.w;
                                qy[g] = vy.x; qy[g + 1] = vy.y; qy[g + 2] = vy.z; qy[g + 3] = vy.w;
                                qz[g] = vz.x; qz[g + 1] = vz.y; qz[g + 2] = vz.z; qz[g + 3] = vz.w;
                            }
#pragma unroll
                            for (int b = 0; b < B; ++b)
#pragma unroll
                                for (int p = 0; p < P; ++p)
                                    m = __builtin_fmaxf(m, __builtin_fmaf(wx[p], qx[b], __builtin_fmaf(wy[p], qy[b], __builtin_fmaf(wz[p], qz[b], -1.0f))));
                            if (__builtin_amdgcn_ballot_w64(m >= 0.0f) == 0ull) continue;
#pragma unroll
                            for (int b = 0; b < B; ++b) {
                                uint32_t slots = 0u;
#pragma unroll
                                for (int p = 0; p < P; ++p) {
                                    const bool cand = __builtin_fmaf(wx[p], qx[b], __builtin_fmaf(wy[p], qy[b], __builtin_fmaf(wz[p], qz[b], -1.0f))) >= 0.0f;
                                    if (__builtin_amdgcn_ballot_w64(cand) != 0ull) slots |= 1u << p;
                                }
                                if (slots == 0u) continue;
#ifdef RT_MASK_STATS
                                if (lane == 0) atomicAdd(A.tile_ctr + 6, (uint32_t)__builtin_popcount(slots));
#endif
                                exact(s_s[s0 + b], slots);
                            }
                            open = need & ~occ;
                            if (__builtin_amdgcn_ballot_w64(open != 0u) == 0ull) {
#ifdef RT_MASK_STATS
                                if (lane == 0) atomicAdd(A.tile_ctr + 7, 1u);
#endif
                                return false;  // early-out: every shadow ray of this wave is blocked
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        count = 0;
                        return true;
                    };
                    bool going = true;
                    for (int cb = 0; cb < n_cl && going; cb += 256) {
                        // ---- cluster vote: four clusters per lane, their loads in flight together; open clusters -> s_cl ----
                        float4 ca[4], cs[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int c = cb + 64 * i + lane;
                            ca[i] = c < n_cl ? clA[c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            cs[i] = c < n_cl ? clB[c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        }
                        int n_open = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int c = cb + 64 * i + lane;
                            const float dot = __builtin_fmaf(ax, ca[i].x, __builtin_fmaf(ay, ca[i].y, az * ca[i].z));
                            const float rhs = __builtin_fmaf(cosT, cs[i].x, -(sinT * cs[i].y));
                            const bool pass = c < n_cl && (wide || dot >= rhs - 4.0e-6f) && !(ca[i].w > sdmax);
                            const uint64_t mask = __builtin_amdgcn_ballot_w64(pass);
                            if (pass) s_cl[n_open + (int)lanes_below(mask)] = c;
                            n_open += __builtin_popcountll(mask);
                        }
#ifdef RT_MASK_STATS
                        if (lane == 0) atomicAdd(A.tile_ctr + 4, (uint32_t)n_open);
#endif
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        // ---- member vote: four open clusters at a time, 16 lanes each ----
                        for (int base = 0; base < n_open && going; base += 4) {
                            const int ci = base + (lane >> 4);
                            const int j = (ci < n_open ? s_cl[ci] : 0) * 16 + (lane & 15);
                            const float4 q = ci < n_open ? lf[j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            const float4 sp = A.sph_o[j];  // the sphere itself, same spatial order: candidates are tested out of LDS, no dependent load
                            const float dot = __builtin_fmaf(ax, q.x, __builtin_fmaf(ay, q.y, az * q.z));
                            const float rhs = __builtin_fmaf(cosT, kConeP1, -(sinT * q.w));
                            const bool entry = q.x != 0.0f || q.y != 0.0f || q.z != 0.0f;  // all zero: padding / always-list / plane / no cluster
                            const bool keep = entry && (wide || dot >= rhs - 8.0e-6f);
                            const uint64_t km = __builtin_amdgcn_ballot_w64(keep);
                            if (km == 0ull) continue;
                            if (keep) {
                                const int at = count + (int)lanes_below(km);
                                s_x[at] = q.x;
                                s_y[at] = q.y;
                                s_z[at] = q.z;
                                s_s[at] = sp;
                            }
                            count += __builtin_popcountll(km);
#ifdef RT_MASK_STATS
                            if (lane == 0) atomicAdd(A.tile_ctr + 5, (uint32_t)__builtin_popcountll(km));
#endif
                            if (count > CAP - 64) going = walk();
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (going && count > 0) (void)walk();
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p)
                if ((occ >> p) & 1u) atomicOr(A.rec_occ + (size_t)(l >> 5) * A.rec_cap + rec[p], 1u << (l & 31));
        }
    }
    if (lane == 0) {  // re-arm the queue (see render_kernel); here every WAVE does exactly one failing fetch
        if (atomicAdd(A.tile_ctr + 1, 1u) == gridDim.x * kWaves - 1u) {
            atomicExch(A.tile_ctr, 0u);
            atomicExch(A.tile_ctr + 1, 0u);
        }
    }
}

// Closest hit (renderer.rs:47-54) of the P bounce rays per lane of ONE WAVE over the whole scene -> best_t / best_i; `need` selects the
// live rays.  No workgroup barrier: the wave works out of its own LDS (s_g: c'x | c'y | c'z | Kc planes of CAP entries, s_s: the
// spheres themselves, s_i: their element indices, s_cl: open clusters).
// Hierarchy, as for the shadow rays: the wave's records are neighbours in the spatial order, so its rays start inside a small ball
// (centre oc, radius ro) and — the reference reflects about the normal at the OLD origin (renderer.rs:68), one normal per sphere on
// the first bounce — point into a narrow cone (axis a, half-angle theta).  A ray of that bundle that the reference's test can accept
// for sphere (c, r) passes, on its forward side, within  rho_e = sqrt(r^2 (1 + 1e-4) + E) + 5.66e-3 |q|  of c (general_filter's margins
// read backwards: delta_ref >= 0 needs dist(c, line)^2 <= rho + 3.06e-5 |q|^2 + 116 u S2, E = 160 u S2max covers the last term), or
// starts inside that radius.  With v = c - oc, D = |v| and T = rho_e + ro this needs  D <= T  or  angle(a, v) <= theta + asin(T / D),
// i.e.  a.v >= cos(theta) sqrt(D^2 - T^2) - sin(theta) T.  Clusters of 16 spatially adjacent spheres are voted first with their
// bounding sphere (A.gcl: centre, radius of the centres + largest member radius), then their members; what survives is listed in LDS
// and walked with the 7-FMA filter, candidates are tested exactly out of LDS.  Every comparison is written so that a NaN or an
// overflow keeps the sphere; bundles wider than 60 degrees keep everything (= the walk over the whole list).
template <int P, int CAP>
__device__ __forceinline__ void trace_bundle(const RenderArgs& A, float* s_g, float4* s_s, int* s_i, int* s_cl, int lane, uint32_t need,
                                             const float (&ox)[P], const float (&oy)[P], const float (&oz)[P], const float (&ux)[P],
                                             const float (&uy)[P], const float (&uz)[P], float (&best_t)[P], int (&best_i)[P])
{
    constexpr int B = kGenBlock;
    static_assert(CAP % B == 0 && CAP >= 128 && B % 4 == 0, "list geometry");
    float fx[P], fy[P], fz[P], gx[P], gy[P], gz[P], a0[P], o2[P], px[P], py[P], pz[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const bool on = (need >> p) & 1u;
        // dead rays get an inert filter: direction 0 and an origin 1e18 away => h = -T << O2, never a candidate of a finite sphere
        const float sx = on ? ux[p] * kGenDirScale : 0.0f, sy = on ? uy[p] * kGenDirScale : 0.0f, sz = on ? uz[p] * kGenDirScale : 0.0f;
        px[p] = on ? ox[p] - A.ball[0] : 1.0e18f;
        py[p] = on ? oy[p] - A.ball[1] : 1.0e18f;
        pz[p] = on ? oz[p] - A.ball[2] : 1.0e18f;
        a0[p] = __builtin_fmaf(sx, px[p], __builtin_fmaf(sy, py[p], sz * pz[p]));
        o2[p] = __builtin_fmaf(px[p], px[p], __builtin_fmaf(py[p], py[p], pz[p] * pz[p])) * kGenO2Scale;
        fx[p] = -sx; fy[p] = -sy; fz[p] = -sz;
        gx[p] = -2.0f * px[p]; gy[p] = -2.0f * py[p]; gz[p] = -2.0f * pz[p];
        best_i[p] = -1;
        best_t[p] = 0.0f;
    }
    for (int a = 0; a < A.n_planes; ++a) {  // planes: no filter, exact test for every live ray
        const int k = A.planes[a];
        const float4 p0 = A.sph_r[k], n = A.pln[k];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float t;
            const bool h = hit_plane_exact(ox[p], oy[p], oz[p], ux[p], uy[p], uz[p], p0, n, t) && ((need >> p) & 1u);
            take_closest(h, t, k, best_t[p], best_i[p]);
        }
    }
    // ---- the bundle: origin ball and direction cone of the live rays with finite data (the others miss every exact test) ----
    uint32_t finite_origin = 0u, fin;
#pragma unroll
    for (int p = 0; p < P; ++p)
        if (((need >> p) & 1u) && __builtin_isfinite(px[p]) && __builtin_isfinite(py[p]) && __builtin_isfinite(pz[p])) finite_origin |= 1u << p;
    float ax, ay, az, sinT, cosT;
    const bool narrow = wave_cone<P>(finite_origin, ux, uy, uz, fin, ax, ay, az, sinT, cosT);
    float sox = 0.0f, soy = 0.0f, soz = 0.0f, cnt = 0.0f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (!((fin >> p) & 1u)) continue;
        sox = sox + px[p]; soy = soy + py[p]; soz = soz + pz[p];
        cnt = cnt + 1.0f;
    }
    cnt = wave_sum(cnt);
    if (!(cnt > 0.0f)) return;  // (wave-uniform)
    const float ocx = wave_sum(sox) / cnt, ocy = wave_sum(soy) / cnt, ocz = wave_sum(soz) / cnt;
    float ro = 0.0f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (!((fin >> p) & 1u)) continue;
        const float ex = px[p] - ocx, ey = py[p] - ocy, ez = pz[p] - ocz;
        ro = __builtin_fmaxf(ro, sqrtf(ex * ex + ey * ey + ez * ez));
    }
    ro = wave_max(ro) * (1.0f + 1.0e-5f) + 1.0e-30f;
    const bool wide = !(narrow && ro < INFINITY);
    const float on_ = sqrtf(ocx * ocx + ocy * ocy + ocz * ocz) + ro;  // >= |o'| of every ray of the bundle
    const float E = 9.5367431640625e-06f * (on_ * on_ + A.ball[3] * 1.0021f);  // 160 u (|o'|^2 + |c'|^2): A.ball[3] = 0.998 Rs^2 >= ... |c'|^2 <= Rs^2
    const float sqrtE = sqrtf(E);
    // cull(v, D2, rho_e): true only if NO ray of the bundle can have an accepted hit on a sphere centred at oc + v with effective radius rho_e
    auto cull = [&](float vx, float vy, float vz, float D2, float D, float rho_e) -> bool {
        const float T = (rho_e + ro) * (1.0f + 1.0e-4f) + 1.0e-4f * D;
        const float w1 = sqrtf(__builtin_fmaxf(0.0f, D2 * (1.0f - 1.0e-5f) - T * T));
        const float dot = ax * vx + ay * vy + az * vz;
        return !wide && D > T && dot < cosT * w1 - sinT * T - 1.0e-5f * D;
    };
    int count = 0;
    float* const s_x = s_g;
    float* const s_y = s_g + CAP;
    float* const s_z = s_g + 2 * CAP;
    float* const s_k = s_g + 3 * CAP;
    auto walk = [&]() {
        const int padded = (count + B - 1) / B * B;
        if (count + lane < padded) {  // pad the last block with entries that never pass (Kc = +inf: h = -inf)
            s_x[count + lane] = 0.0f;
            s_y[count + lane] = 0.0f;
            s_z[count + lane] = 0.0f;
            s_k[count + lane] = INFINITY;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int s0 = 0; s0 < padded; s0 += B) {
            float bx[B], by[B], bz[B], bw[B];
#pragma unroll
            for (int g = 0; g < B; g += 4) {  // wave-uniform addresses: broadcast ds_read_b128, four spheres per read
                const float4 vx = *reinterpret_cast<const float4*>(&s_x[s0 + g]);
                const float4 vy = *reinterpret_cast<const float4*>(&s_y[s0 + g]);
                const float4 vz = *reinterpret_cast<const float4*>(&s_z[s0 + g]);
                const float4 vw = *reinterpret_cast<const float4*>(&s_k[s0 + g]);
                bx[g] = vx.x; bx[g + 1] = vx.y; bx[g + 2] = vx.z; bx[g + 3] = vx.w;
                by[g] = vy.x; by[g + 1] = vy.y; by[g + 2] = vy.z; by[g + 3] = vy.w;
                bz[g] = vz.x; bz[g + 1] = vz.y; bz[g + 2] = vz.z; bz[g + 3] = vz.w;
                bw[g] = vw.x; bw[g + 1] = vw.y; bw[g + 2] = vw.z; bw[g + 3] = vw.w;
            }
            float m[P];
            bool any = false;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                m[p] = -INFINITY;
#pragma unroll
                for (int b = 0; b < B; ++b) m[p] = __builtin_fmaxf(m[p], general_filter<P>(p, gx, gy, gz, fx, fy, fz, a0, bx[b], by[b], bz[b], bw[b]));
                any = any || m[p] >= o2[p];
            }
            if (__builtin_amdgcn_ballot_w64(any) == 0ull) continue;
            // some lane has a candidate in this block (every bounce ray starts ON a sphere): per-sphere vote from the registers the block
            // was evaluated from, exact test out of LDS and only for the ray slots that hold a candidate
#pragma unroll
            for (int b = 0; b < B; ++b) {
                uint32_t slots = 0u;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const bool cand = general_filter<P>(p, gx, gy, gz, fx, fy, fz, a0, bx[b], by[b], bz[b], bw[b]) >= o2[p];
                    if (__builtin_amdgcn_ballot_w64(cand) != 0ull) slots |= 1u << p;
                }
                if (slots == 0u) continue;
                const float4 sr = s_s[s0 + b];
                const int k = s_i[s0 + b];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (!((slots >> p) & 1u)) continue;  // wave-uniform: none of the 64 rays of this slot is a candidate of sphere k
                    float t;
                    const bool h = hit_general_exact(ox[p], oy[p], oz[p], ux[p], uy[p], uz[p], sr, t) && ((need >> p) & 1u);
                    take_closest(h, t, k, best_t[p], best_i[p]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        count = 0;
    };
    const int n_cl = A.n_pad / 16;
    for (int cb = 0; cb < n_cl; cb += 256) {
        // ---- cluster vote: four clusters per lane, their loads in flight together; open clusters -> s_cl ----
        float4 cc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cb + 64 * i + lane;
            cc[i] = c < n_cl ? A.gcl[c] : make_float4(0.0f, 0.0f, 0.0f, -1.0f);
        }
        int n_open = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cb + 64 * i + lane;
            const float vx = cc[i].x - ocx, vy = cc[i].y - ocy, vz = cc[i].z - ocz;
            const float D2 = vx * vx + vy * vy + vz * vz, D = sqrtf(D2);
            const float rho_e = cc[i].w + sqrtE + 5.66e-3f * (D + cc[i].w + ro);  // w = radius of the centres + largest member radius (1 + 1e-4)
            const bool open_ = c < n_cl && !(cc[i].w < 0.0f) && !cull(vx, vy, vz, D2, D, rho_e);  // w = -1: empty cluster
            const uint64_t mask = __builtin_amdgcn_ballot_w64(open_);
            if (open_) s_cl[n_open + (int)lanes_below(mask)] = c;
            n_open += __builtin_popcountll(mask);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- member vote: four open clusters at a time, 16 lanes each ----
        for (int base = 0; base < n_open; base += 4) {
            const int ci = base + (lane >> 4);
            const int j = (ci < n_open ? s_cl[ci] : 0) * 16 + (lane & 15);
            const float4 g = A.gen_o[j];
            const float4 sp = A.sph_o[j];
            const float vx = g.x - ocx, vy = g.y - ocy, vz = g.z - ocz;
            const float D2 = vx * vx + vy * vy + vz * vz, D = sqrtf(D2);
            const float rho_e = sqrtf(sp.w * sp.w * (1.0f + 1.0e-4f) + E) + 5.66e-3f * (D + ro);
            // Kc = +inf: padding / plane, never a candidate; Kc = -inf: a sphere outside the filter's argument, always a candidate
            const bool keep = ci < n_open && g.w != INFINITY && (g.w == -INFINITY || !cull(vx, vy, vz, D2, D, rho_e));
            const uint64_t km = __builtin_amdgcn_ballot_w64(keep);
            if (km == 0ull) continue;
            if (keep) {
                const int at = count + (int)lanes_below(km);
                s_x[at] = g.x;
                s_y[at] = g.y;
                s_z[at] = g.z;
                s_k[at] = g.w;
                s_s[at] = sp;
                s_i[at] = A.order[j];
            }
            count += __builtin_popcountll(km);
            if (count > CAP - 64) walk();
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (count > 0) walk();
}

// One iteration (A.iter) of renderer.rs:46-73 for the live records: hit point, lights in list order (occlusion bits from
// shadow_mask_kernel when RT_SHADOWS), (x) reflectiveness; then either the final store (last iteration) or the bounce (normal at
// the OLD origin, renderer.rs:68; reflect, object/mod.rs:71-73; no epsilon) and the closest hit of the bounce ray (trace_bundle).
// A ray that misses keeps its colour for good (renderer.rs:47-56: prev_info stays untouched), so its pixel is stored at once and the
// record is marked dead (rec_idx = -1).  A wave owns 64*P consecutive entries of the record list — of the spatially sorted list of
// live records (A.rec_perm) when the iteration was sorted (A.sorted: always when it traces bounces or casts shadow rays), so that its
// bounce rays form the narrow bundle trace_bundle feeds on; of the emission order otherwise (neighbouring pixels).
template <int P, int CAP>
__device__ __forceinline__ void shade_records(const RenderArgs& A, float* s_g, float4* s_s, int* s_i, int* s_cl, int lane, uint32_t count, uint32_t wave_first)
{
    const bool shadows = (A.flags & RT_SHADOWS) != 0u;
    const bool last = A.iter + 1u >= A.depth;  // renderer.rs:67 (i < limit)
    const size_t cap = A.rec_cap;

    float dx[P], dy[P], dz[P], ox[P], oy[P], oz[P], hx[P], hy[P], hz[P], cr[P], cg[P], cb[P], lr[P], lg[P], lb[P], best_t[P];
    int best_i[P];
    uint32_t pix[P], rec[P];
    uint32_t alive = 0u;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const uint32_t e = wave_first + p * 64 + lane;
        bool ok = e < count;
        const uint32_t r = A.sorted ? (ok ? A.rec_perm[e] : 0u) : e;
        rec[p] = r;
        best_i[p] = ok ? A.rec_idx[r] : -1;
        ok = ok && best_i[p] >= 0;
        pix[p] = ok ? A.rec_pix[r] : 0u;
        best_t[p] = ok ? A.rec_t[r] : 0.0f;
        if (ok) alive |= 1u << p;
        load_ray(A, r, ok, pix[p], ox[p], oy[p], oz[p], dx[p], dy[p], dz[p]);
        if (A.iter == 0u) {
            cr[p] = cg[p] = cb[p] = 0.0f;  // renderer.rs:42
        } else {
            const size_t i = ok ? r : 0u;
            cr[p] = A.rec_state[kStCr * cap + i]; cg[p] = A.rec_state[kStCg * cap + i]; cb[p] = A.rec_state[kStCb * cap + i];
        }
        hx[p] = ox[p] + best_t[p] * dx[p];  // renderer.rs:57
        hy[p] = oy[p] + best_t[p] * dy[p];
        hz[p] = oz[p] + best_t[p] * dz[p];
        lr[p] = cr[p];                      // renderer.rs:59
        lg[p] = cg[p];
        lb[p] = cb[p];
    }
    for (int l = 0; l < A.n_lights; ++l) {  // renderer.rs:60-64, list order
        const float4 lp = A.lpos[l];
        const float4 lc = A.lcol[l];
        if (lc.w != 0.0f) {  // light/ambient.rs:23-25
#pragma unroll
            for (int p = 0; p < P; ++p) {
                lr[p] = lr[p] + lc.x;
                lg[p] = lg[p] + lc.y;
                lb[p] = lb[p] + lc.z;
            }
            continue;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {  // light/point.rs:25-30
            const float vx = hx[p] - lp.x, vy = hy[p] - lp.y, vz = hz[p] - lp.z;
            const float dist = sqrtf(dot3_ref(vx, vy, vz, vx, vy, vz));
            const float in = lp.w / (dist * dist);
            bool some = !(in <= 1.1920929e-7f);
            if (shadows && ((alive >> p) & 1u)) some = some && !((A.rec_occ[(size_t)(l >> 5) * cap + rec[p]] >> (l & 31)) & 1u);
            if (some) {
                lr[p] = lr[p] + in * lc.x;
                lg[p] = lg[p] + in * lc.y;
                lb[p] = lb[p] + in * lc.z;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const bool on = (alive >> p) & 1u;
        const float4 rf = A.refl[on ? best_i[p] : 0];
        cr[p] = lr[p] * rf.x;  // renderer.rs:66
        cg[p] = lg[p] * rf.y;
        cb[p] = lb[p] * rf.z;
    }
    if (last) {
#pragma unroll
        for (int p = 0; p < P; ++p)
            if ((alive >> p) & 1u) store_pixel(A, pix[p], cr[p], cg[p], cb[p]);
        return;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const bool on = (alive >> p) & 1u;
        const float4 s = A.sph_r[on ? best_i[p] : 0];
        float nx = ox[p] - s.x, ny = oy[p] - s.y, nz = oz[p] - s.z;  // renderer.rs:68 — normal at the OLD origin
        unit3(nx, ny, nz);
        if (A.n_planes) {  // a plane's normal is the same everywhere (extension)
            const float4 pn = A.pln[on ? best_i[p] : 0];
            if (pn.w != 0.0f) { nx = pn.x; ny = pn.y; nz = pn.z; }
        }
        const float dn = dot3_ref(dx[p], dy[p], dz[p], nx, ny, nz);  // object/mod.rs:72
        const float two_dn = 2.0f * dn;
        float ex = dx[p] - two_dn * nx, ey = dy[p] - two_dn * ny, ez = dz[p] - two_dn * nz;
        unit3(ex, ey, ez);
        dx[p] = ex; dy[p] = ey; dz[p] = ez;
        ox[p] = hx[p]; oy[p] = hy[p]; oz[p] = hz[p];  // renderer.rs:70
    }
    trace_bundle<P, CAP>(A, s_g, s_s, s_i, s_cl, lane, alive, ox, oy, oz, dx, dy, dz, best_t, best_i);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (!((alive >> p) & 1u)) continue;
        const uint32_t r = rec[p];
        A.rec_idx[r] = best_i[p];
        if (best_i[p] < 0) {
            store_pixel(A, pix[p], cr[p], cg[p], cb[p]);  // every later iteration misses identically: the colour is final
        } else {
            A.rec_t[r] = best_t[p];
            A.rec_state[kStOx * cap + r] = ox[p]; A.rec_state[kStOy * cap + r] = oy[p]; A.rec_state[kStOz * cap + r] = oz[p];
            A.rec_state[kStDx * cap + r] = dx[p]; A.rec_state[kStDy * cap + r] = dy[p]; A.rec_state[kStDz * cap + r] = dz[p];
            A.rec_state[kStCr * cap + r] = cr[p]; A.rec_state[kStCg * cap + r] = cg[p]; A.rec_state[kStCb * cap + r] = cb[p];
        }
    }
}

// The four waves of a workgroup are independent (no barrier): each owns 64 * P entries and its own slice of LDS.  (A persistent grid
// whose waves pull their units from a queue was measured and lost, 0.66 against 0.31 ms on C3 depth 2 when the trace still walked the
// whole list: returning atomics on one address complete at only ~40-80 M/s chip-wide; profiles/r03_tuning.md.)
template <int P, int CAP>
__global__ __launch_bounds__(kWaves * 64) void shade_iter_kernel(const RenderArgs A)
{
    __shared__ __attribute__((aligned(16))) float s_g[kWaves][4 * CAP];
    __shared__ float4 s_s[kWaves][CAP];
    __shared__ int s_i[kWaves][CAP];
    __shared__ int s_cl[kWaves][256];
    const int wave = threadIdx.x >> 6;
    const uint32_t count = A.sorted ? A.rec_count[1] : A.rec_count[0];
    const uint32_t wave_first = (blockIdx.x * kWaves + (uint32_t)wave) * (64 * P);
    if (wave_first >= count) return;
    shade_records<P, CAP>(A, s_g[wave], s_s[wave], s_i[wave], s_cl[wave], threadIdx.x & 63, count, wave_first);
}

// Zeroes scratch words with a KERNEL, not hipMemsetAsync: `planes` planes of `n` words, `stride` words apart, starting at p; plus up to
// 16 words at q (the tile-queue counters / the record count).  Kernel launches are captured and replayed faithfully by HIP graphs;
// memset nodes on this scratch were not (replays of a captured RT_SHADOWS launch kept stale occlusion bits), and callers under graph
// capture are what rt_render_rows_device_noalloc is for.
__global__ __launch_bounds__(256) void clear_words_kernel(uint32_t* p, size_t n, size_t stride, uint32_t planes, uint32_t* q, uint32_t nq)
{
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (uint32_t pl = 0; pl < planes; ++pl)
        for (size_t i = i0; i < n; i += step) p[(size_t)pl * stride + i] = 0u;
    if (i0 < nq) q[i0] = 0u;
}

inline hipError_t clear_words(uint32_t* p, size_t n, size_t stride, uint32_t planes, uint32_t* q, uint32_t nq, hipStream_t stream)
{
    const size_t want = (n + 255) / 256;
    const uint32_t grid = (uint32_t)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(clear_words_kernel, dim3(grid), dim3(256), 0, stream, p, n, stride, planes, q, nq);
    return hipGetLastError();
}

inline bool is_general(const RenderArgs& a) { return a.depth > 1u || (a.flags & RT_SHADOWS) != 0u; }

#ifndef RT_B
#define RT_B 16
#endif

// Unused dynamic LDS requested at launch to cap the resident workgroups per CU of the general-ray kernels (sweep in profiles/r02_tuning.md)
#ifndef RT_MASK_LDS_PAD
#define RT_MASK_LDS_PAD 0
#endif
#ifndef RT_SHADE_LDS_PAD
#define RT_SHADE_LDS_PAD 0
#endif
constexpr uint32_t kMaskLdsPad = RT_MASK_LDS_PAD, kShadeLdsPad = RT_SHADE_LDS_PAD;
constexpr int kB = RT_B;       // spheres per wave vote in the primary loop (sweep in profiles/r01_tuning.md)
#ifndef RT_SHADE_P
#define RT_SHADE_P 2
#endif
constexpr int kShadeP = RT_SHADE_P;  // records per lane in shade_iter_kernel
#ifndef RT_SHADE_CAP
#define RT_SHADE_CAP 128
#endif
constexpr int kShadeCap = RT_SHADE_CAP;  // entries of the per-wave sphere list of trace_bundle (36 B each in LDS)
#ifndef RT_MASK_P
#define RT_MASK_P 4
#endif
constexpr int kMaskP = RT_MASK_P;   // shadow rays per lane in shadow_mask_kernel
#ifndef RT_MASK_TB
#define RT_MASK_TB 4
#endif
constexpr int kMaskTB = RT_MASK_TB; // tiles a wave takes from the queue per atomic

}  // namespace

bool render_needs_records(const RenderArgs& args) { return is_general(args); }

TilePlan plan_tiles(uint32_t w, uint32_t rows, uint32_t slots)
{
    TilePlan tp;
    tp.strips = (w + kWaves * kLX - 1) / (kWaves * kLX);
    const uint64_t tiles16 = (uint64_t)tp.strips * ((rows + 15) / 16);
    if (tiles16 * 8 <= (uint64_t)slots * 5) {
        // a very small launch (well under one 16-row tile per resident workgroup): 8-row tiles fill the chip better
        // (135 rows of a 4K frame alone on a stream: 0.050 vs 0.065 ms)
        tp.rows4 = 0;
        tp.rows2 = (rows + 7) / 8 * 8;
    } else if (tiles16 < slots) {
        // up to one 16-row tile per resident workgroup (the band of one of 8 ranks at 4K): 16-row tiles only, the efficient shape
        // (one broadcast LDS read per 16 FMAs).  Such launches are what a multi-GPU host keeps several of in flight, and then the
        // next launch covers the tail of this one: three streams, 270 rows 73.9 -> 68.3 us per frame against 8-row tiles; alone on
        // a stream the two time the same.  From one tile per workgroup on the sectioned queue wins in that pipeline (540 rows:
        // 140.6 against 143.2 us per step), although a plain grid would be 3 % faster alone on a stream (profiles/r01_tuning.md)
        tp.rows4 = (rows + 15) / 16 * 16;
        tp.rows2 = 0;
    } else {
        // 16-row tiles for 98 % of the band, then 2 % in 8-row tiles and what rounding leaves over in 4-row tiles: the smaller shapes
        // are less efficient per pixel (fewer FMAs per LDS read), so only the very end of the launch gets them (sweep in
        // profiles/r01_tuning.md: C3 -1.2 %, C4 -2.8 % against the earlier 85 / 10 / 5 split; all 16-row tiles lose 2 %)
        tp.rows4 = (uint32_t)((uint64_t)rows * 980 / 1000) / 16 * 16;
        tp.rows2 = (uint32_t)((uint64_t)rows * 20 / 1000) / 8 * 8;
    }
    const uint32_t rest = rows > tp.rows4 + tp.rows2 ? rows - tp.rows4 - tp.rows2 : 0;
    tp.tiles4 = tp.strips * (tp.rows4 / 16);
    tp.tiles2 = tp.strips * (tp.rows2 / 8);
    tp.tiles1 = tp.strips * ((rest + 3) / 4);
    return tp;
}

namespace {

// Resident workgroup slots of the current device for a persistent kernel (CUs x workgroups of 256 threads per CU), cached per
// (kernel, device) in `cache` (one array per kernel; a benign race: concurrent callers compute the same value).
template <typename Kernel>
hipError_t resident_slots(Kernel kernel, uint32_t (&cache)[64], uint32_t& slots, uint32_t dyn_lds = 0)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (cache[dev] == 0) {
        int cus = 0, per_cu = 0;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kWaves * 64, dyn_lds)) != hipSuccess) return e;
        if (cus <= 0 || per_cu <= 0) return hipErrorInvalidValue;
        cache[dev] = (uint32_t)cus * (uint32_t)per_cu;
    }
    slots = cache[dev];
    return hipSuccess;
}

template <bool EMIT>
hipError_t primary_slots(uint32_t& slots)
{
    static uint32_t cache[64] = {};
    return resident_slots(render_kernel<kB, EMIT>, cache, slots);
}

#ifndef RT_CULL_CAP
#define RT_CULL_CAP 128
#endif
constexpr int kCullCap = RT_CULL_CAP;  // entries of the per-wave sphere list of render_cull_kernel (16 B each in LDS)

// RT_CULL: a plain grid of 64 x 16 pixel tiles, one per workgroup.
template <bool EMIT>
hipError_t launch_primary_cull(RenderArgs a, hipStream_t stream)
{
    if (!a.pfilt_o || !a.porder || !a.pclA || !a.pclB) return hipErrorInvalidValue;
    a.strips = (a.w + kWaves * kLX - 1) / (kWaves * kLX);
    const uint64_t n_tiles = (uint64_t)a.strips * ((a.y1 - a.y0 + kCullP * kLY - 1) / (kCullP * kLY));
    if (n_tiles >= (1ull << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((render_cull_kernel<EMIT, kCullCap>), dim3((uint32_t)n_tiles), dim3(kWaves * 64), 0, stream, a);
    return hipGetLastError();
}

hipError_t mask_slots_of_device(uint32_t& slots)
{
    static uint32_t cache[64] = {};
    return resident_slots(shadow_mask_kernel<kMaskP, kShadowB, kShadowCap, kMaskTB>, cache, slots, kMaskLdsPad);
}

// Cut the band into the three tile sections of render_kernel and launch the persistent grid.
template <bool EMIT>
hipError_t launch_primary(RenderArgs a, hipStream_t stream)
{
    if (!a.tile_ctr) return hipErrorInvalidValue;
    uint32_t slots = 0;
    hipError_t e = primary_slots<EMIT>(slots);
    if (e != hipSuccess) return e;
    const TilePlan tp = plan_tiles(a.w, a.y1 - a.y0, slots);
    a.strips = tp.strips;
    a.rows4 = tp.rows4;
    a.rows2 = tp.rows2;
    a.tiles4 = tp.tiles4;
    a.tiles2 = tp.tiles2;
    a.tiles1 = tp.tiles1;
    const uint32_t n_tiles = a.tiles4 + a.tiles2 + a.tiles1;
    // A launch of equal tiles (the two small-band plans above) goes out as a plain grid, one tile per workgroup: the queue has
    // nothing to balance there, and workgroups that leave as they finish let the RCCL kernels of a multi-GPU step in (band pipeline
    // with a real RCCL kernel per frame: 84 vs 87 us per step; 135-row band alone 0.044 vs 0.051 ms).
    a.static_grid = (a.tiles4 == 0 || (a.tiles2 == 0 && a.tiles1 == 0)) ? 1u : 0u;
    const uint32_t grid = (a.static_grid || n_tiles < slots) ? n_tiles : slots;
    hipLaunchKernelGGL((render_kernel<kB, EMIT>), dim3(grid), dim3(kWaves * 64), 0, stream, a);
    return hipGetLastError();
}
}  // namespace

// The resident-slot caches of the three persistent kernels for the current device (launch_primary / launch_render fill them on first
// use otherwise): called at context creation so that no render call does set-up work.
hipError_t prime_launch_caches()
{
    uint32_t slots = 0;
    hipError_t e;
    if ((e = primary_slots<false>(slots)) != hipSuccess) return e;
    if ((e = primary_slots<true>(slots)) != hipSuccess) return e;
    return mask_slots_of_device(slots);
}

hipError_t launch_render(const RenderArgs& args, hipStream_t stream, PhaseEvents* phases)
{
    if (phases) phases->iters = 0;
    if (args.y1 <= args.y0 || args.w == 0) return hipSuccess;
    const bool cull = (args.flags & RT_CULL) != 0u;
    if (!is_general(args)) return cull ? launch_primary_cull<false>(args, stream) : launch_primary<false>(args, stream);

    const bool shadows = (args.flags & RT_SHADOWS) != 0u && args.n_lights > 0;
    const uint64_t npix = (uint64_t)(args.y1 - args.y0) * args.w;
    if (!args.rec_count || !args.rec_pix || !args.rec_idx || !args.rec_t || args.rec_cap < npix) return hipErrorInvalidValue;
    if ((args.depth > 1u && !args.rec_state) || (shadows && !args.rec_occ)) return hipErrorInvalidValue;
    if ((shadows || args.depth > 1u) && (!args.rec_perm || !args.rec_hit || !args.sort_bins)) return hipErrorInvalidValue;
    // RT_CULL: the emit kernel also takes the histogram of the first record sort (finish_tile), so its bins are cleared up front
    const bool emit_hist = cull && args.sort_bins != nullptr;
    hipError_t e = emit_hist ? clear_words(args.sort_bins, (size_t)args.n + 1, 0, 1, args.rec_count, 1, stream) : clear_words(nullptr, 0, 0, 0, args.rec_count, 1, stream);
    if (e != hipSuccess) return e;
    if (phases && (e = hipEventRecord(phases->ev[0], stream)) != hipSuccess) return e;
    if ((e = cull ? launch_primary_cull<true>(args, stream) : launch_primary<true>(args, stream)) != hipSuccess) return e;
    if (phases && (e = hipEventRecord(phases->ev[1], stream)) != hipSuccess) return e;

    // The record count lives on the device: the grids below are sized for the worst case (every pixel of the band hit) or persistent,
    // and workgroups with nothing to do leave at once — no host round trip between the launches.
    uint32_t mask_slots = 0;
    if (shadows && (e = mask_slots_of_device(mask_slots)) != hipSuccess) return e;
    RenderArgs a = args;
    const uint32_t shade_wgs = (uint32_t)((npix + kWaves * 64 * kShadeP - 1) / (kWaves * 64 * kShadeP));
    const uint32_t occ_words = ((uint32_t)args.n_lights + 31u) / 32u;
    for (uint32_t it = 0; it < args.depth; ++it) {  // renderer.rs:46
        a.iter = it;
        const bool tracing = it + 1u < args.depth;  // this iteration ends with a bounce
        a.sorted = (shadows || tracing) ? 1u : 0u;
        if (a.sorted) {
            // spatial sort of the live records (rec_idx of this iteration) -> rec_perm, their hit points -> rec_hit, live count -> rec_count[1]
            const bool have_hist = it == 0 && emit_hist;  // the culled emit kernel took the histogram of iteration 0 itself
            if (!have_hist && (e = clear_words(a.sort_bins, (size_t)a.n + 1, 0, 1, nullptr, 0, stream)) != hipSuccess) return e;
            const uint32_t sort_wgs = (uint32_t)(npix + 255) / 256 < 2048u ? (uint32_t)(npix + 255) / 256 : 2048u;
            if (!have_hist) hipLaunchKernelGGL((sort_records_kernel<false>), dim3(sort_wgs), dim3(256), 0, stream, a);
            hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(256), 0, stream, a);
            hipLaunchKernelGGL((sort_records_kernel<true>), dim3(sort_wgs), dim3(256), 0, stream, a);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        if (shadows) {
            // only the words of this band's records (at most npix per plane), not the whole capacity of the scratch
            if ((e = clear_words(a.rec_occ, (size_t)npix, a.rec_cap, occ_words, nullptr, 0, stream)) != hipSuccess) return e;
            const uint64_t worst_tiles = (npix + 64 * kMaskP - 1) / (64 * kMaskP) * (uint64_t)args.n_lights;
            if (worst_tiles >= (1ull << 32)) return hipErrorInvalidValue;  // the tile queue counts in 32 bits
            const uint64_t worst_wgs = (worst_tiles + kWaves - 1) / kWaves;
            const uint32_t grid = worst_wgs < mask_slots ? (uint32_t)worst_wgs : mask_slots;
            hipLaunchKernelGGL((shadow_mask_kernel<kMaskP, kShadowB, kShadowCap, kMaskTB>), dim3(grid), dim3(kWaves * 64), kMaskLdsPad, stream, a);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        const bool timed = phases && it < kPhaseIters;
        if (timed && (e = hipEventRecord(phases->ev[2 + 2 * it], stream)) != hipSuccess) return e;
        hipLaunchKernelGGL((shade_iter_kernel<kShadeP, kShadeCap>), dim3(shade_wgs), dim3(kWaves * 64), kShadeLdsPad, stream, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (timed) {
            if ((e = hipEventRecord(phases->ev[3 + 2 * it], stream)) != hipSuccess) return e;
            phases->iters = it + 1;
        }
    }
    return hipSuccess;
}

void render_scratch_layout(const RenderArgs& args, bool& need_state, uint32_t& occ_words, uint32_t& sort_planes, uint32_t& bin_words)
{
    need_state = args.depth > 1u;
    occ_words = ((args.flags & RT_SHADOWS) != 0u && args.n_lights > 0) ? ((uint32_t)args.n_lights + 31u) / 32u : 0u;
    const bool sorted = need_state || occ_words > 0u;  // iterations that trace bounces or cast shadow rays work on spatially sorted records
    sort_planes = sorted ? 4u : 0u;                    // rec_perm + three planes of hit points
    bin_words = sorted ? ((uint32_t)args.n + 1u + 3u) / 4u * 4u : 0u;
}

const char* render_kernel_name(const RenderArgs& args)
{
    if (!is_general(args)) return (args.flags & RT_CULL) ? "render_cull_kernel<primary>" : "render_kernel<primary>";
    return (args.flags & RT_SHADOWS) ? "render_kernel<emit> + shadow_mask_kernel + shade_iter_kernel" : "render_kernel<emit> + shade_iter_kernel";
}

}  // namespace rtamd
