// render_kernels.h — launch interface of the gfx950 render kernels (device code in render_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rtamd {

struct RenderArgs {
    float m[12];  // rows 0..2 of the row-major camera Mat4 (row 3 only feeds w, which vec3.rs:702-707 drops)
    uint32_t w, h;    // full frame size (ray formula uses global pixel indices, display.rs:99)
    uint32_t y0, y1;  // row band rendered by this launch
    uint32_t depth;   // renderer.rs:32
    uint32_t flags;   // RT_SHADOWS | RT_OUT_RGB8 | RT_CULL
    int32_t n, n_pad, n_always, n_lights;
    const float4* filt;     // n_pad
    const float4* sph;      // n_pad (cx,cy,cz,k)
    const float4* sph_r;    // n_pad (cx,cy,cz,radius)
    const float4* gen_o;    // n_pad, SPATIAL order: (c'x, c'y, c'z, Kc) general-origin filter constants (c' = c - ball centre)
    const float4* gcl;      // n_pad / 16: per cluster of 16 consecutive positions (centre c', radius of the centres + largest member radius; -1 = empty)
    const float4* refl;     // n_pad
    const int32_t* always;  // n_always
    const float4* lpos;     // n_lights
    const float4* lcol;     // n_lights
    const float4* lfilt;        // n_lights * n_pad: per-light shadow filter constants in SPATIAL order, w = P2 of the member test
    const float4* lclA;         // n_lights * n_pad / 16: per cluster of 16 entries (cone axis x, y, z, dmin)
    const float4* lclB;         // n_lights * n_pad / 16: (cos, sin of the cluster's half-angle, 0, 0)
    const int32_t* order;       // n_pad: element at spatial position j, -1 = padding
    const float4* sph_o;        // n_pad: sph_r in spatial order (position j holds element order[j]; padding: NaN radius)
    const int32_t* rank;        // n_pad: spatial position of element i (sort key of the hit records)
    const int32_t* lalways;     // per-light always-test lists, concatenated
    const int32_t* lalways_off; // n_lights + 1
    const int32_t* porder;      // RT_CULL, n_pad: element at position j of the primary rays' direction order, -1 = no entry
    const float4* pfilt_o;      // RT_CULL, n_pad: filt in that order, w = P2 of the member test; all zero = no entry
    const float4* pclA;         // RT_CULL, n_pad / 16: per cluster of 16 positions (cone axis x, y, z, 0)
    const float4* pclB;         // RT_CULL, n_pad / 16: (cos, sin of the cluster's half-angle, 0, 0)
    const float4* pln;          // n_pad: (nx, ny, nz, 1) for RT_PLANE elements (extension), 0 for spheres
    const int32_t* planes;      // n_planes indices of the plane elements (tested exactly for every ray)
    int32_t n_planes;
    float ball[4];              // hit points farther than sqrt(ball[3]) from ball[0..2] bypass the per-light shadow filter
    uint8_t* out;           // band rows, 4 or 3 B/pixel
    int32_t* hit_idx;       // nullable
    // dense hit records (library-owned scratch, only for depth > 1 or RT_SHADOWS): one counter + capacity >= band pixels
    uint32_t* rec_count;  // [0] = records emitted by the primary kernel, [1] = live records of the current iteration (sort_scan_kernel)
    uint32_t* rec_pix;
    int32_t* rec_idx;
    float* rec_t;
    uint32_t rec_cap;     // capacity (records) of every rec_* plane
    uint32_t iter;        // iteration of renderer.rs:46 the general-ray kernels work on (set by launch_render)
    uint32_t sorted;      // this iteration's rec_perm / rec_hit / rec_count[1] are valid (set by launch_render)
    float* rec_state;     // depth > 1: 9 planes of rec_cap floats (origin, direction, colour of the bounce ray); else null
    uint32_t* rec_occ;    // RT_SHADOWS: ceil(n_lights / 32) planes of rec_cap words, occlusion bit per (record, light); else null
    uint32_t* rec_perm;   // RT_SHADOWS or depth > 1: rec_cap words, the live records of this iteration in spatial order (rec_count[1] entries); else null
    float* rec_hit;       // RT_SHADOWS or depth > 1: 3 planes of rec_cap floats, the hit points of those records in the same (sorted) order; else null
    uint32_t* sort_bins;  // RT_SHADOWS or depth > 1: n + 1 words, histogram / running offsets of the record sort; else null
    // tile queue of the persistent primary kernel: tile_ctr[0] = next tile, tile_ctr[1] = workgroups that have left (library-owned,
    // zero between launches); the rest is filled in by launch_render
    uint32_t* tile_ctr;
    uint32_t strips;                  // 64-pixel column strips per tile row
    uint32_t tiles4, tiles2, tiles1;  // number of 16-, 8- and 4-row tiles, in queue order
    uint32_t rows4, rows2;            // rows covered by the first two sections
    uint32_t static_grid;             // 1: plain grid, workgroup b renders tile b only (small bands); 0: persistent grid + tile queue
};

// One tile-queue counter slot (A.tile_ctr): word 0 = next tile, word 1 = workgroups (waves) that have left; 64 B apart.
constexpr uint32_t kTileSlotWords = 16;

// How launch_render cuts a band of `rows` rows x `w` columns into the tile queue of the persistent primary kernel, given the number of
// resident workgroup slots: 64-pixel strips; the first rows4 rows in 16-row tiles, the next rows2 in 8-row tiles, the rest in 4-row tiles
// (rows past the band's end are masked in the kernel).  Host-only; exposed for the CPU tests (rt_debug_tile_plan).
struct TilePlan {
    uint32_t strips, rows4, rows2, tiles4, tiles2, tiles1;
};
TilePlan plan_tiles(uint32_t w, uint32_t rows, uint32_t slots);

// Optional timing of the phases of a general launch (depth > 1 or RT_SHADOWS): ev[0] before and ev[1] after the primary (emit) kernel,
// then per iteration it < kPhaseIters of renderer.rs:46: ev[2 + 2 it] after shadow_mask_kernel (and the clear of its occlusion words),
// ev[3 + 2 it] after shade_iter_kernel.  `iters` = iterations recorded.  Events are the caller's (timing enabled).
constexpr uint32_t kPhaseIters = 4;
struct PhaseEvents {
    hipEvent_t ev[2 + 2 * kPhaseIters];
    uint32_t iters;
};

// Enqueue the render of args.[y0,y1) on `stream`. Returns the HIP status of the launch.  `phases` (nullable) is filled for general launches.
hipError_t launch_render(const RenderArgs& args, hipStream_t stream, PhaseEvents* phases = nullptr);

// Looks up, for the current device, what launch_render would otherwise look up on its first call (resident workgroup slots of the
// persistent kernels).  Called once per context.
hipError_t prime_launch_caches();

// True if launch_render needs the rec_* scratch buffers for these args (depth > 1 or RT_SHADOWS).
bool render_needs_records(const RenderArgs& args);

// Which optional scratch these args need beside count/pix/idx/t: rec_state (9 float planes), rec_occ (occ_words planes), sort_planes
// more planes (rec_perm, rec_hit x 3) and bin_words words of sort bins.
void render_scratch_layout(const RenderArgs& args, bool& need_state, uint32_t& occ_words, uint32_t& sort_planes, uint32_t& bin_words);

// Name of the kernel variant launch_render picks for these args (for profiles / bench output).
const char* render_kernel_name(const RenderArgs& args);

// Flops executed per primary-ray/sphere filter test by the hot loop (for roofline accounting).
constexpr int kFilterFlopsPerTest = 6;  // three v_fma_f32

}  // namespace rtamd
