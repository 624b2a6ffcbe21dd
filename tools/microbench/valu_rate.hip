// VALU issue-rate microbenchmark (GPU box): how many cycles does one SIMD of gfx950 spend per
// non-fused f32 VALU wave-instruction at the occupancies k_trace_shade runs at?  Decides the
// denominator of bench.py's "valu" roofline.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));

// Packed f32: v_pk_fma_f32 (PK 0), v_pk_mul_f32 + v_pk_add_f32 (PK 1) — two lane-operations per lane and instruction.
template <int PK>
__global__ __launch_bounds__(256) void k_rate_pk(float* out, int iters, float a_, float b_) {
    const f2 a = {a_, a_}, b = {b_, b_};
    f2 x[8];
    for (int k = 0; k < 8; ++k) x[k] = f2{threadIdx.x * 1e-3f + k, threadIdx.x * 2e-3f + k};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (PK == 0) x[k] = __builtin_elementwise_fma(x[k], a, b);
                else x[k] = x[k] * a + b;
            }
        }
    }
    f2 s = {0.0f, 0.0f};
    for (int k = 0; k < 8; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {  // mul + add, unfused (two VALU instructions per pair), 8 independent chains
                x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
                x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b;
            } else if (KIND == 1) {  // explicit fma
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            } else {  // compare + select (v_cmp + v_cndmask) and a mul
                x0 = (x0 > b ? x1 : x0) * a; x1 = (x1 > b ? x2 : x1) * a; x2 = (x2 > b ? x3 : x2) * a; x3 = (x3 > b ? x4 : x3) * a;
                x4 = (x4 > b ? x5 : x4) * a; x5 = (x5 > b ? x6 : x5) * a; x6 = (x6 > b ? x7 : x6) * a; x7 = (x7 > b ? x0 : x7) * a;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND>
void launch(int blocks, float* d_out, int iters) {
    if (KIND >= 10) k_rate_pk<KIND - 10><<<blocks, 256>>>(d_out, iters, 1.0001f, 1e-7f);
    else k_rate<KIND><<<blocks, 256>>>(d_out, iters, 1.0001f, 1e-7f);
}

template <int KIND>
void run(const char* name, int waves_per_simd, int valu_per_inner, float* d_out) {
    const int blocks = 256 * waves_per_simd, iters = 4000;  // one block = 4 waves = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch<KIND>(blocks, d_out, 10);
    hipEventRecord(e0);
    launch<KIND>(blocks, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)waves_per_simd * iters * 8 * valu_per_inner;  // wave-instructions one SIMD issued
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double cycles = ms * 1e-3 * clk_khz * 1e3;
    printf("%-22s waves/SIMD %d  %.3f ms  %.2f cycles per VALU wave-instruction per SIMD (at %d MHz)  %.1f T lane-ops/s\n", name,
           waves_per_simd, ms, cycles / insts_per_simd, clk_khz / 1000, insts_per_simd * 1024 * 64 / (ms * 1e-3) / 1e12);
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 6}) {
        run<0>("mul+add (unfused)", w, 16, d_out);
        run<1>("fma", w, 8, d_out);
        run<2>("cmp+cndmask+mul", w, 24, d_out);
        run<10>("v_pk_fma_f32", w, 8, d_out);
        run<11>("v_pk_mul + v_pk_add", w, 16, d_out);
    }
    hipFree(d_out);
    return 0;
}
