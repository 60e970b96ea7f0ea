// Microbenchmark 3: v_fma_f32 issue rate against waves per SIMD (1..8) and unroll depth, to see whether any occupancy reaches the
// nominal 2 cycles per wave64 instruction.  Reports wave-instructions per ns per SIMD (wall clock) and cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int U, int MODE>
__global__ __launch_bounds__(256) void k_fma(float* __restrict__ out, const float* __restrict__ in, int iters, unsigned long long* __restrict__ clk)
{
    const int tid = threadIdx.x;
    float x = in[tid & 63], y = in[(tid + 7) & 63], z = in[(tid + 3) & 63];
    float a[U];
#pragma unroll
    for (int i = 0; i < U; ++i) a[i] = in[(tid + i) & 63];
    // every load must have landed before the timed loop: otherwise the compiler sinks one s_waitcnt per accumulator into the loop body
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < U; ++i) asm volatile("" : "+v"(a[i]));
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
    if (MODE == 2) __builtin_amdgcn_s_setprio(3);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (MODE == 5) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));            // VOP2, 4-byte encoding
            else if (MODE == 6) asm volatile("v_fma_f32 %0, %1, %2, -1.0" : "=v"(a[i]) : "v"(x), "v"(y));   // VOP3 with the inline constant, 8 bytes
            else if (MODE == 7) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));     // VOP2
            else if (MODE == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            else if (MODE == 4) {   // the render loop's mix: 6 FMAs per max3
                if (i % 7 == 6) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[i - 1]), "v"(a[i - 2]));
                else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
            } else if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(x), "v"(y), "v"(z));   // no accumulator dependence
            else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
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

static void run(const char* name, kern_t k, int unroll, int w, int threads, float* d_out, float* d_in, unsigned long long* d_clk, int ncu, long total)
{
    const int waves_per_wg = threads / 64;
    const int blocks = ncu * 4 * w / waves_per_wg;        // w waves per SIMD
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
    printf("%-12s U=%-3d wg=%-3d w=%d ms=%7.2f  inst/ns/SIMD=%6.3f  sclk=%5.3f GHz  cycles/inst=%5.2f  TFLOP/s=%6.1f\n", name, unroll, threads, w, ms, rate, g,
           g / rate, rate * ncu * 4 * 128 / 1e3);
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
    const long total = 3200000;   // wave-instructions per SIMD per launch / 1: a few ms
    for (int w : {1, 2, 3, 4, 5, 6, 8}) run("acc", k_fma<16, 0>, 16, w, 256, d_out, d_in, d_clk, ncu, total);
    for (int w : {1, 2, 4, 8}) run("acc", k_fma<64, 0>, 64, w, 256, d_out, d_in, d_clk, ncu, total);
    for (int w : {1, 2, 4, 8}) run("indep", k_fma<16, 1>, 16, w, 256, d_out, d_in, d_clk, ncu, total);
    for (int w : {1, 2, 4, 8}) run("acc_wg64", k_fma<16, 0>, 16, w, 64, d_out, d_in, d_clk, ncu, total);
    for (int w : {2, 4, 8}) run("acc_prio3", k_fma<16, 2>, 16, w, 256, d_out, d_in, d_clk, ncu, total);
    for (int w : {1, 2, 4, 8}) run("max3", k_fma<16, 3>, 16, w, 256, d_out, d_in, d_clk, ncu, total);
    for (int w : {2, 4, 8}) run("fmac_e32", k_fma<64, 5>, 64, w, 256, d_out, d_in, d_clk, ncu, total);
    for (int w : {2, 4, 8}) run("fma_vop3_c", k_fma<64, 6>, 64, w, 256, d_out, d_in, d_clk, ncu, total);
    for (int w : {2, 4, 8}) run("mul_e32", k_fma<64, 7>, 64, w, 256, d_out, d_in, d_clk, ncu, total);
    for (int w : {2, 4, 8}) run("fma_vop3", k_fma<64, 0>, 64, w, 256, d_out, d_in, d_clk, ncu, total);
    for (int w : {1, 2, 3, 4, 8}) run("mix6:1", k_fma<56, 4>, 56, w, 256, d_out, d_in, d_clk, ncu, total);
    return 0;
}
