// Microbenchmark 2: issue cost of single VALU instructions on gfx950, 4 and 8 waves per SIMD, long enough to reach steady clocks.
// Reports wave-instructions per ns per SIMD (wall clock) and the effective shader clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define DEF(NAME, ASM)                                                                                                     \
    __global__ __launch_bounds__(256) void k_##NAME(float* __restrict__ out, const float* __restrict__ in, int iters,      \
                                                    unsigned long long* __restrict__ clk)                                  \
    {                                                                                                                      \
        const int tid = threadIdx.x;                                                                                       \
        float x = in[tid & 63], y = in[(tid + 7) & 63];                                                                    \
        float a[16];                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = in[(tid + i) & 63];                                          \
        float sy = __builtin_amdgcn_readfirstlane(in[65]);                                                                 \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                       \
        for (int it = 0; it < iters; ++it) {                                                                               \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(x), "v"(y), "s"(sy) : "vcc", "s40", "s41"); \
        }                                                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                       \
        float acc = 0.f;                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) acc += a[i];                                                        \
        out[blockIdx.x * blockDim.x + tid] = acc;                                                                          \
        if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }                                \
    }

DEF(fma, "v_fma_f32 %0, %1, %2, %0")
DEF(fma_s, "v_fma_f32 %0, %1, %3, %0")
DEF(fma_const, "v_fma_f32 %0, %1, %2, -1.0")
DEF(fma_clamp, "v_fma_f32 %0, %1, %2, %0 clamp")
DEF(fmac, "v_fmac_f32 %0, %1, %2")
DEF(mul, "v_mul_f32 %0, %1, %0")
DEF(add, "v_add_f32 %0, %1, %0")
DEF(sub, "v_sub_f32 %0, %1, %0")
DEF(maxf, "v_max_f32 %0, %1, %0")
DEF(max3, "v_max3_f32 %0, %0, %1, %2")
DEF(med3, "v_med3_f32 %0, %0, %1, %2")
DEF(or3, "v_or3_b32 %0, %0, %1, %2")
DEF(and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF(or2, "v_or_b32 %0, %1, %0")
DEF(and2, "v_and_b32 %0, %1, %0")
DEF(bfi, "v_bfi_b32 %0, %0, %1, %2")
DEF(perm, "v_perm_b32 %0, %0, %1, %2")
DEF(lshl_or, "v_lshl_or_b32 %0, %0, 1, %2")
DEF(mov, "v_mov_b32 %0, %1")
DEF(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF(cmp_vcc, "v_cmp_ge_f32 vcc, %0, %1")
DEF(cmp_sgpr, "v_cmp_ge_f32 s[40:41], %0, %1")
DEF(cvt_u32, "v_cvt_u32_f32 %0, %0")
DEF(sqrt, "v_sqrt_f32 %0, %0")
DEF(rcp, "v_rcp_f32 %0, %0")
DEF(add_u32, "v_add_u32 %0, %1, %0")
DEF(xor3, "v_xor_b32 %0, %1, %0")
DEF(min_u32, "v_min_u32 %0, %1, %0")
DEF(max_i32, "v_max_i32 %0, %1, %0")
DEF(max3_i32, "v_max3_i32 %0, %0, %1, %2")

typedef void (*kern_t)(float*, const float*, int, unsigned long long*);

static void run(const char* name, kern_t k, int w, float* d_out, float* d_in, unsigned long long* d_clk, int ncu, int iters)
{
    const int blocks = ncu * w;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<<<blocks, 256>>>(d_out, d_in, iters / 10, d_clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<<<blocks, 256>>>(d_out, d_in, iters, d_clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> clk(blocks * 2);
    CHECK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz(blocks);
    for (int b = 0; b < blocks; ++b) ghz[b] = (double)clk[2 * b] / (double)clk[2 * b + 1] * 0.1;  // 100 MHz realtime
    std::sort(ghz.begin(), ghz.end());
    double rate = (double)blocks * 4 * iters * 16 / (ms * 1e6) / (ncu * 4);  // wave-insts per ns per SIMD
    double g = ghz[blocks / 2];
    printf("%-10s w=%d ms=%7.2f  inst/ns/SIMD=%6.3f  sclk=%5.3f GHz  cycles/inst=%5.2f\n", name, w, ms, rate, g, g / rate);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    float* d_in; float* d_out; unsigned long long* d_clk;
    std::vector<float> h(256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.25f + 0.5f * (float)((i * 2654435761u) % 1000) / 1000.f;
    CHECK(hipMalloc(&d_in, h.size() * 4));
    CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)ncu * 8 * 256 * 4));
    CHECK(hipMalloc(&d_clk, (size_t)ncu * 8 * 2 * 8));
#define RUN(NAME) for (int w : {4, 8}) run(#NAME, k_##NAME, w, d_out, d_in, d_clk, ncu, 200000 / w);
    RUN(fma) RUN(fma_s) RUN(fma_const) RUN(fma_clamp) RUN(fmac) RUN(mul) RUN(add) RUN(sub) RUN(maxf) RUN(max3) RUN(med3) RUN(or3) RUN(and_or)
    RUN(or2) RUN(and2) RUN(bfi) RUN(perm) RUN(lshl_or) RUN(mov) RUN(cndmask) RUN(cmp_vcc) RUN(cmp_sgpr)
    RUN(cvt_u32) RUN(sqrt) RUN(rcp) RUN(add_u32) RUN(xor3) RUN(min_u32) RUN(max_i32) RUN(max3_i32)
    return 0;
}
