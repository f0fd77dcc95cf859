// ubench_valu.hip -- VALU issue-rate microbenchmark for gfx950 (MI355X).
//
// Measures, per SIMD, the cycles per wave64 instruction of the ops the all-pairs inner loop is
// made of (v_fma_f32, v_pk_fma_f32, v_sqrt_f32, v_rcp_f32, v_rsq_f32 and the real mix), at 1, 2,
// 4 and 8 waves per SIMD.  The issue-bound ceiling quoted in DESIGN.md comes from this table.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kIters = 4000;
constexpr int kChains = 16;  // independent accumulators per lane
constexpr int kRep = 4;      // chain sweeps per loop iteration

enum Op { FMA, PKFMA, SQRT, RCP, RSQ, MIX_FMA_RCP, PAIR };

template <int OP>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, float seed) {
    float a[kChains];
    v2f p[kChains];
    const float b = seed * 0.999f, c = seed * 1e-3f;
    const v2f b2{b, b}, c2{c, c};
#pragma unroll
    for (int i = 0; i < kChains; ++i) {
        a[i] = seed + i + threadIdx.x * 1e-3f;
        p[i] = v2f{a[i], a[i] + 0.5f};
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < kRep; ++r) {
#pragma unroll
            for (int i = 0; i < kChains; ++i) {
                if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == PKFMA)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(c2));
                if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (OP == RSQ) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
                if (OP == MIX_FMA_RCP) {  // 6 fma : 1 rcp, roughly the loop's ratio
                    if (i % 7 == 6)
                        asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                    else
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < kChains; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char *name, float *out, unsigned long long *cyc, int n_cu) {
    printf("%-12s", name);
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
        const int threads = wps >= 4 ? 1024 : 256 * wps;  // 4 SIMDs per CU
        const int blocks_per_cu = wps >= 4 ? wps / 4 : 1;
        const int grid = n_cu * blocks_per_cu;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        // warm up for ~150 ms of back-to-back launches: the chip's clock needs ~50-100 ms under
        // load to settle (profiles/r01_warmup.txt), and the costs below should be steady-state
        for (int w = 0; w < 40 * (wps >= 4 ? 1 : 4 / wps); ++w)
            hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, out, cyc, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, out, cyc, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const int n_waves = grid * threads / 64;
        std::vector<unsigned long long> h(n_waves);
        CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * n_waves, hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= n_waves;
        const double instr = (double)kIters * kRep * kChains;
        // two clocks: wall (HIP events) and the shader-cycle counter s_memtime.  A wave's own
        // elapsed cycles / its instruction count / wps = issue interval per SIMD.
        const double ns_per_instr_simd = (double)ms * 1e6 / (instr * wps);
        printf("  wps=%d: %.3f ns (%.2f cyc by s_memtime)", wps, ns_per_instr_simd,
               mean / (instr * wps));
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clock %d kHz\n", prop.name, n_cu, prop.clockRate);
    float *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&out, sizeof(float) * n_cu * 2 * 1024));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * n_cu * 2 * 16));
    run<FMA>("v_fma_f32", out, cyc, n_cu);
    run<PKFMA>("v_pk_fma_f32", out, cyc, n_cu);
    run<SQRT>("v_sqrt_f32", out, cyc, n_cu);
    run<RCP>("v_rcp_f32", out, cyc, n_cu);
    run<RSQ>("v_rsq_f32", out, cyc, n_cu);
    run<MIX_FMA_RCP>("6fma:1rcp", out, cyc, n_cu);
    return 0;
}
