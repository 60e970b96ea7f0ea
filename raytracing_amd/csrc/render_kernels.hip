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

// EXTENSION RT_SHADOWS: is the segment p -> lp blocked? `need` lanes take part; wave-level early-out once every
// needing lane is occluded. Exact general-origin test, so the CPU oracle's any-hit answer is reproduced bit for bit.
__device__ __forceinline__ bool occluded(const RenderArgs& A, bool need, float px, float py, float pz, float lx, float ly, float lz)
{
    float vx = lx - px, vy = ly - py, vz = lz - pz;
    float dist = sqrtf(dot3_ref(vx, vy, vz, vx, vy, vz));
    float dx = vx / dist, dy = vy / dist, dz = vz / dist;
    bool occ = false;
    for (int k = 0; k < A.n; ++k) {
        if (__builtin_amdgcn_ballot_w64(need && !occ) == 0ull) break;
        float t;
        bool h = hit_general_exact(px, py, pz, dx, dy, dz, A.sph_r[k], t);
        occ = occ || (need && h && t > RT_SHADOW_EPS && t < dist);
    }
    return occ;
}

// renderer.rs:59-66 — c = color + sum(lights(p)) in list order, then (x) reflectiveness of element k
__device__ __forceinline__ void shade(const RenderArgs& A, bool active, int k, float px, float py, float pz, float& cr, float& cg, float& cb)
{
    const bool shadows = (A.flags & RT_SHADOWS) != 0u;
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
            bool some = !(in <= 1.1920929e-7f);
            if (shadows) {
                bool occ = occluded(A, active && some, px, py, pz, lp.x, lp.y, lp.z);
                some = some && !occ;
            }
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

// Closest hit for a general-origin ray (bounces): exact brute force over the list in order.
__device__ __forceinline__ void closest_general(const RenderArgs& A, bool active, float ox, float oy, float oz, float dx, float dy, float dz, float& best_t, int& best_i)
{
    best_i = -1;
    best_t = 0.0f;
    for (int k = 0; k < A.n; ++k) {
        float t;
        bool h = hit_general_exact(ox, oy, oz, dx, dy, dz, A.sph_r[k], t);
        take_closest(active && h, t, k, best_t, best_i);
    }
}

template <int PX, int PY, int B, bool GENERAL>
__global__ __launch_bounds__(kWaves * 64) void render_kernel(const RenderArgs A)
{
    constexpr int P = PX * PY;
    // filter constants of the current chunk, SoA so that four spheres' x (y, z) come back in one broadcast ds_read_b128
    __shared__ __attribute__((aligned(16))) float s_x[kChunk], s_y[kChunk], s_z[kChunk];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int lx = lane & (kLX - 1);
    const int ly = lane >> 4;
    const uint32_t x_base = (blockIdx.x * kWaves + wave) * (kLX * PX) + lx;
    const uint32_t y_base = A.y0 + blockIdx.y * (kLY * PY) + ly;

    float dx[P], dy[P], dz[P], best_t[P];
    int best_i[P];
    uint32_t inb = 0;

    // ---- primary rays: display.rs:98-100, renderer.rs:43 ----
    const float fw = (float)A.w, fh = (float)A.h;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        uint32_t x = x_base + kLX * (p % PX);
        uint32_t y = y_base + kLY * (p / PX);
        if (x < A.w && y < A.y1) inb |= 1u << p;
        x = min(x, A.w - 1u);
        y = min(y, A.y1 - 1u);
        float v0 = 2.0f * ((float)x / fw) - 1.0f;
        float v1 = 2.0f * ((float)y / fh) - 1.0f;
        float v2 = 2.0f * (1.0f / 1.0f) - 1.0f;
        float v3 = v2;
        float qx = dot4_ref(A.m + 0, v0, v1, v2, v3);
        float qy = dot4_ref(A.m + 4, v0, v1, v2, v3);
        float qz = dot4_ref(A.m + 8, v0, v1, v2, v3);
        unit3(qx, qy, qz);
        dx[p] = qx;
        dy[p] = qy;
        dz[p] = qz;
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

    // ---- hot loop: conservative 3-FMA filter over the whole list, exact test only for candidate blocks ----
    for (int c0 = 0; c0 < A.n_pad; c0 += kChunk) {
        const int cn = min(kChunk, A.n_pad - c0);
        __syncthreads();  // every wave is done reading the previous chunk
        for (int i = tid; i < cn; i += kWaves * 64) {
            const float4 q = A.filt[c0 + i];
            s_x[i] = q.x;
            s_y[i] = q.y;
            s_z[i] = q.z;
        }
        __syncthreads();
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
                // rare: some lane has a candidate in this block. Find the spheres, run the reference's exact test.
                for (int b = 0; b < B; ++b) {
                    const float4 q = A.filt[c0 + s0 + b];
                    bool cand = false;
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        float f = __builtin_fmaf(dx[p], q.x, __builtin_fmaf(dy[p], q.y, __builtin_fmaf(dz[p], q.z, -1.0f)));
                        cand = cand || (f >= 0.0f);
                    }
                    if (__builtin_amdgcn_ballot_w64(cand) == 0ull) continue;
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

    // ---- shading, bounces, store ----
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const bool in_frame = (inb >> p) & 1u;
        const uint32_t x = x_base + kLX * (p % PX);
        const uint32_t y = y_base + kLY * (p / PX);
        float cr = 0.0f, cg = 0.0f, cb = 0.0f;
        int k = best_i[p];
        float t = best_t[p];
        const int first_hit = k;
        float ox = 0.0f, oy = 0.0f, oz = 0.0f;
        float rdx = dx[p], rdy = dy[p], rdz = dz[p];
        if (!GENERAL) {
            const bool hit = k >= 0;
            // renderer.rs:57, object/mod.rs:61-63 — origin + t * direction
            float px = ox + t * rdx, py = oy + t * rdy, pz = oz + t * rdz;
            float r = 0.0f, g = 0.0f, b = 0.0f;
            shade(A, hit, k, px, py, pz, r, g, b);
            if (hit) { cr = r; cg = g; cb = b; }
        } else {
            bool alive = true;
            for (uint32_t it = 0; it < A.depth; ++it) {
                if (it > 0) closest_general(A, alive, ox, oy, oz, rdx, rdy, rdz, t, k);
                alive = alive && (k >= 0);  // a miss leaves prev_info untouched: all later iterations miss identically
                if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;
                float px = ox + t * rdx, py = oy + t * rdy, pz = oz + t * rdz;
                float r = cr, g = cg, b = cb;  // renderer.rs:59
                shade(A, alive, k, px, py, pz, r, g, b);
                if (alive) { cr = r; cg = g; cb = b; }
                if (it + 1u < A.depth) {  // renderer.rs:67 (i < limit)
                    const float4 s = A.sph_r[alive ? k : 0];
                    float nx = ox - s.x, ny = oy - s.y, nz = oz - s.z;  // renderer.rs:68 — normal at the OLD origin
                    unit3(nx, ny, nz);
                    float dn = dot3_ref(rdx, rdy, rdz, nx, ny, nz);  // object/mod.rs:72
                    float two_dn = 2.0f * dn;
                    float ex = rdx - two_dn * nx, ey = rdy - two_dn * ny, ez = rdz - two_dn * nz;
                    unit3(ex, ey, ez);
                    if (alive) {
                        rdx = ex; rdy = ey; rdz = ez;
                        ox = px; oy = py; oz = pz;
                    }
                }
            }
        }
        if (in_frame) {
            const size_t pix = (size_t)(y - A.y0) * A.w + x;
            const uint32_t r8 = quantise(cr), g8 = quantise(cg), b8 = quantise(cb);
            if (A.flags & RT_OUT_RGB8) {
                uint8_t* o = A.out + pix * 3;
                o[0] = (uint8_t)r8;
                o[1] = (uint8_t)g8;
                o[2] = (uint8_t)b8;
            } else {
                reinterpret_cast<uint32_t*>(A.out)[pix] = r8 | (g8 << 8) | (b8 << 16) | 0xff000000u;
            }
            if (A.hit_idx) A.hit_idx[pix] = first_hit;
        }
    }
}

template <int PX, int PY, int B, bool GENERAL>
hipError_t launch_variant(const RenderArgs& args, hipStream_t stream)
{
    const uint32_t rows = args.y1 - args.y0;
    dim3 grid((args.w + kWaves * kLX * PX - 1) / (kWaves * kLX * PX), (rows + kLY * PY - 1) / (kLY * PY));
    hipLaunchKernelGGL((render_kernel<PX, PY, B, GENERAL>), grid, dim3(kWaves * 64), 0, stream, args);
    return hipGetLastError();
}

inline bool is_general(const RenderArgs& a) { return a.depth > 1u || (a.flags & RT_SHADOWS) != 0u; }

}  // namespace

// Geometry chosen by the sweep recorded in profiles/r01_tuning.md: 16x4 lanes, 1x4 pixels per lane (16x16-pixel wave tile),
// one vote per 8 spheres.
hipError_t launch_render(const RenderArgs& args, hipStream_t stream)
{
    if (args.y1 <= args.y0 || args.w == 0) return hipSuccess;
    if (is_general(args)) return launch_variant<1, 4, 8, true>(args, stream);
    return launch_variant<1, 4, 8, false>(args, stream);
}

const char* render_kernel_name(const RenderArgs& args) { return is_general(args) ? "render_kernel<1,4,8,general>" : "render_kernel<1,4,8,primary>"; }

}  // namespace rtamd
