// Microbenchmark: per-SIMD issue rate of the VALU/LDS/SMEM instructions the render kernel is built from.
// Build: hipcc -O2 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run on the GPU box.
// Reports cycles per wave-instruction per SIMD (shader clock, s_memtime) and chip TFLOP/s by wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

enum Mode { FMA_V = 0, FMA_S, PKFMA_V, PKFMA_S, MUL_V, PKMUL_V, MAX3_V, CMP_V, MIX_RT, MIX_RT_PK, LDS_B128_FMA, SLOAD_FMA, FMA_V_DEP, NMODES };
static const char* mode_name[NMODES] = {"v_fma_f32 vvv", "v_fma_f32 v,s,v", "v_pk_fma_f32 vvv", "v_pk_fma_f32 v,s,v", "v_mul_f32", "v_pk_mul_f32", "v_max3_f32",
  "v_cmp_ge_f32", "mix: mul+3fma+.5max3 (SGPR sphere)", "mix pk: pkmul+3pkfma+max3 (8 tests/20 inst)", "ds_read_b128 bcast + 16 fma", "s_load_dwordx4 + 16 fma", "v_fma_f32 dependent chain"};
// VALU instructions per loop iteration and flops per lane per iteration
static const int mode_insts[NMODES] = {16, 16, 16, 16, 16, 16, 16, 16, 18, 20, 16, 16, 16};
static const int mode_flops[NMODES] = {32, 32, 64, 64, 16, 32, 0, 0, 28, 56, 32, 32, 32};

template <int MODE>
__global__ __launch_bounds__(256) void bench(float* __restrict__ out, const float* __restrict__ in, int iters, unsigned long long* __restrict__ cyc)
{
    __shared__ f4 lds[256];
    const int tid = threadIdx.x;
    float x = in[tid & 63], y = in[(tid + 7) & 63];
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = in[(tid + i) & 63];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    f2 px = {x, y};
    // wave-uniform scalars
    float sx = in[64], sy = in[65];
    sx = __builtin_amdgcn_readfirstlane(sx);
    sy = __builtin_amdgcn_readfirstlane(sy);
    f2 ps = {sx, sy};
    lds[tid] = f4{x, y, sx, sy};
    __syncthreads();
    const f4* gsph = reinterpret_cast<const f4*>(in + 128);

    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == FMA_V) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        } else if (MODE == FMA_V_DEP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(x), "v"(y));
        } else if (MODE == FMA_S) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(sy));
        } else if (MODE == PKFMA_V) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(px), "v"(px));
        } else if (MODE == PKFMA_S) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(px), "s"(ps));
        } else if (MODE == MUL_V) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
        } else if (MODE == PKMUL_V) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(px));
        } else if (MODE == MAX3_V) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        } else if (MODE == CMP_V) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
        } else if (MODE == MIX_RT) {
            // 4 pixels (a[0..2],a[3..5],a[6..8],a[9..11] = dx,dy,dz) against one sphere in SGPRs: per pixel mul,fma,fma,fma ; 2 max3
            float d0, d1, d2, d3;
            asm volatile(
                "v_mul_f32 %0, %4, %16\n v_mul_f32 %1, %7, %16\n v_mul_f32 %2, %10, %16\n v_mul_f32 %3, %13, %16\n"
                "v_fma_f32 %0, %5, %17, %0\n v_fma_f32 %1, %8, %17, %1\n v_fma_f32 %2, %11, %17, %2\n v_fma_f32 %3, %14, %17, %3\n"
                "v_fma_f32 %0, %6, %16, %0\n v_fma_f32 %1, %9, %16, %1\n v_fma_f32 %2, %12, %16, %2\n v_fma_f32 %3, %15, %16, %3\n"
                "v_fma_f32 %0, %0, %0, -%17\n v_fma_f32 %1, %1, %1, -%17\n v_fma_f32 %2, %2, %2, -%17\n v_fma_f32 %3, %3, %3, -%17\n"
                : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "s"(sx), "s"(sy));
            asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %0, %0, %3, %4" : "+v"(a[15]) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
        } else if (MODE == MIX_RT_PK) {
            // 8 pixels as 4 packed pairs: p[0..2], p[3..5] (two pairs) + again -> 4 pairs; per pair pk_mul, 3 pk_fma ; pk_max x2 ... 18 insts, 8 tests
            f2 d0, d1, d2, d3;
            asm volatile(
                "v_pk_mul_f32 %0, %4, %10 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %7, %10 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %4, %11 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %7, %11 op_sel_hi:[1,0]\n"
                "v_pk_fma_f32 %0, %5, %11, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %8, %11, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %5, %10, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %8, %10, %3 op_sel_hi:[1,0,1]\n"
                "v_pk_fma_f32 %0, %6, %10, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %9, %10, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %6, %11, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %9, %11, %3 op_sel_hi:[1,0,1]\n"
                "v_pk_fma_f32 %0, %0, %0, %11 op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %1, %1, %1, %11 op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                "v_pk_fma_f32 %2, %2, %2, %10 op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %3, %3, %3, %10 op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
                : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
                : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "s"(ps), "s"(px.x == 123.f ? ps : f2{sy, sx}));
            asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %0, %0, %3, %4\n v_max3_f32 %0, %0, %5, %6\n v_max3_f32 %0, %0, %7, %8" : "+v"(a[15]) : "v"(d0.x), "v"(d0.y), "v"(d1.x), "v"(d1.y), "v"(d2.x), "v"(d2.y), "v"(d3.x), "v"(d3.y));
        } else if (MODE == LDS_B128_FMA) {
            f4 s = lds[(it & 63)];  // wave-uniform address -> broadcast ds_read_b128
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(s[i & 3]), "v"(y));
        } else if (MODE == SLOAD_FMA) {
            f4 s = gsph[it & 255];  // uniform global load -> s_load_dwordx4
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(y), "s"(s[i & 3]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + tid] = acc;
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
static void run(int waves_per_simd, float* d_out, float* d_in, unsigned long long* d_cyc, int ncu)
{
    const int iters = 20000;
    const int blocks = ncu * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    bench<MODE><<<blocks, 256>>>(d_out, d_in, 100, d_cyc);  // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    bench<MODE><<<blocks, 256>>>(d_out, d_in, iters, d_cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(blocks * 4);
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    std::sort(cyc.begin(), cyc.end());
    double med = (double)cyc[cyc.size() / 2];
    double insts = (double)iters * mode_insts[MODE];
    double cyc_per_inst_per_simd = med / (insts * waves_per_simd);  // s_memtime ticks per wave-instruction issued by one SIMD
    double tflops = (double)blocks * 256 * iters * mode_flops[MODE] / (ms * 1e-3) / 1e12;
    double ginst = (double)blocks * 4 * insts / (ms * 1e-3) / 1e9 / (ncu * 4);  // wave-insts per ns per SIMD == GHz-normalised rate
    printf("%-44s w/SIMD=%d  ms=%8.3f  ticks/inst/SIMD=%6.3f  winst/ns/SIMD=%6.3f  TFLOP/s=%7.2f\n", mode_name[MODE], waves_per_simd, ms, cyc_per_inst_per_simd, ginst, tflops);
    fflush(stdout);
}

template <int MODE>
static void sweep(float* d_out, float* d_in, unsigned long long* d_cyc, int ncu)
{
    for (int w : {1, 2, 4, 8}) run<MODE>(w, d_out, d_in, d_cyc, ncu);
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    printf("device %s  CUs=%d  clock=%d kHz\n", prop.name, ncu, prop.clockRate);
    float* d_in; float* d_out; unsigned long long* d_cyc;
    std::vector<float> h(128 + 4 * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.25f + 0.5f * (float)((i * 2654435761u) % 1000) / 1000.f;
    CHECK(hipMalloc(&d_in, h.size() * 4));
    CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)ncu * 8 * 256 * 4));
    CHECK(hipMalloc(&d_cyc, (size_t)ncu * 8 * 4 * 8));
    sweep<FMA_V>(d_out, d_in, d_cyc, ncu);
    sweep<FMA_V_DEP>(d_out, d_in, d_cyc, ncu);
    sweep<FMA_S>(d_out, d_in, d_cyc, ncu);
    sweep<PKFMA_V>(d_out, d_in, d_cyc, ncu);
    sweep<PKFMA_S>(d_out, d_in, d_cyc, ncu);
    sweep<MUL_V>(d_out, d_in, d_cyc, ncu);
    sweep<PKMUL_V>(d_out, d_in, d_cyc, ncu);
    sweep<MAX3_V>(d_out, d_in, d_cyc, ncu);
    sweep<CMP_V>(d_out, d_in, d_cyc, ncu);
    sweep<MIX_RT>(d_out, d_in, d_cyc, ncu);
    sweep<MIX_RT_PK>(d_out, d_in, d_cyc, ncu);
    sweep<LDS_B128_FMA>(d_out, d_in, d_cyc, ncu);
    sweep<SLOAD_FMA>(d_out, d_in, d_cyc, ncu);
    return 0;
}
