// Microbenchmark 4 (round 2): issue rate of candidate FOLD instructions against v_fma_f32, at the render kernel's occupancy (4 waves per
// SIMD) and at 8.  Question: does any instruction that can fold two filter results into an accumulator (v_max3_f32 today) issue at
// the FP32-FMA rate?  Loops are 64 independent accumulators x the instruction under test, nothing else (check with --save-temps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define OPS(X) \
    X(0, "v_fma_f32 %0, %1, %2, %0") \
    X(1, "v_max3_f32 %0, %0, %1, %2") \
    X(2, "v_or3_b32 %0, %0, %1, %2") \
    X(3, "v_and_or_b32 %0, %1, %2, %0") \
    X(4, "v_or_b32_e32 %0, %1, %0") \
    X(5, "v_and_b32_e32 %0, %1, %0") \
    X(6, "v_max_f32_e32 %0, %1, %0") \
    X(7, "v_min_f32_e32 %0, %1, %0") \
    X(8, "v_add_f32_e32 %0, %1, %0") \
    X(9, "v_add_u32_e32 %0, %1, %0") \
    X(10, "v_med3_f32 %0, %0, %1, %2") \
    X(11, "v_add3_u32 %0, %0, %1, %2") \
    X(12, "v_fma_f32 %0, %1, %2, %0 clamp") \
    X(13, "v_max_i32_e32 %0, %1, %0") \
    X(14, "v_min3_f32 %0, %0, %1, %2") \
    X(15, "v_mul_f32_e32 %0, %1, %0") \
    X(16, "v_sub_f32_e32 %0, %1, %0") \
    X(17, "v_lshl_or_b32 %0, %1, 1, %0") \
    X(18, "v_xor_b32_e32 %0, %1, %0") \
    X(19, "v_mad_u32_u24 %0, %1, %2, %0") \
    X(20, "v_fmac_f32_e32 %0, %1, %2") \
    X(21, "v_max_u32_e32 %0, %1, %0") \
    X(22, "v_cndmask_b32_e32 %0, %1, %0, vcc") \
    X(23, "v_bfi_b32 %0, %1, %2, %0") \
    X(24, "v_add_f32_e64 %0, %1, %0 clamp") \
    X(25, "v_mul_f32_e64 %0, %1, %0 clamp")

template <int MODE>
__global__ __launch_bounds__(256) void k_op(float* __restrict__ out, const float* __restrict__ in, int iters, unsigned long long* __restrict__ clk)
{
    constexpr int U = 64;
    const int tid = threadIdx.x;
    float x = in[tid & 63], y = in[(tid + 7) & 63];
    float a[U];
#pragma unroll
    for (int i = 0; i < U; ++i) a[i] = in[(tid + i) & 63];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < U; ++i) asm volatile("" : "+v"(a[i]));
    asm volatile("" : "+v"(x), "+v"(y));
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
#define X(N, S) if (MODE == N) asm volatile(S : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
            OPS(X)
#undef X
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < U; ++i) acc += a[i];
    out[blockIdx.x * blockDim.x + tid] = acc;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

typedef void (*kern_t)(float*, const float*, int, unsigned long long*);

static void run(const char* name, kern_t k, int w, float* d_out, float* d_in, unsigned long long* d_clk, int ncu, long total)
{
    const int unroll = 64, threads = 256, waves_per_wg = 4;
    const int blocks = ncu * 4 * w / waves_per_wg;
    const int iters = (int)(total / w / unroll);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<<<blocks, threads>>>(d_out, d_in, iters / 4, d_clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<<<blocks, threads>>>(d_out, d_in, iters, d_clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> clk(blocks * 2);
    CHECK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz(blocks);
    for (int b = 0; b < blocks; ++b) ghz[b] = (double)clk[2 * b] / (double)clk[2 * b + 1] * 0.1;
    std::sort(ghz.begin(), ghz.end());
    double rate = (double)blocks * waves_per_wg * (double)iters * unroll / (ms * 1e6) / (ncu * 4);
    double g = ghz[blocks / 2];
    printf("%-44s w=%d ms=%7.2f  inst/ns/SIMD=%6.3f  sclk=%5.3f GHz  cycles/inst=%5.2f\n", name, w, ms, rate, g, g / rate);
    fflush(stdout);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
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
    CHECK(hipMalloc(&d_out, (size_t)ncu * 4 * 8 * 64 * 4));
    CHECK(hipMalloc(&d_clk, (size_t)ncu * 4 * 8 * 2 * 8));
    const long total = 2400000;
    for (int w : {4, 8}) {
#define X(N, S) run(S, k_op<N>, w, d_out, d_in, d_clk, ncu, total);
        OPS(X)
#undef X
    }
    return 0;
}
