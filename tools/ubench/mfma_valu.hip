// Does the fp32 MFMA (v_mfma_f32_16x16x4_f32) overlap with independent VALU work of the same wave / of another wave on the
// same SIMD?  Three kernels, one wave per SIMD unless noted:  M: 4 independent MFMA chains;  V: independent v_fma chains;
// MV: both interleaved 1 MFMA : 7 v_fma in program order.  If the matrix pipe were separate, t(MV) ~ max(t(M), t(V)).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_valu.hip -o build/ubench_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters, float a, float b) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float v[28];
    for (int i = 0; i < 28; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // volatile asm: program order is exactly what is written here
            if (MODE & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
            if (MODE & 2) {
#pragma unroll
                for (int j = 0; j < 7; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u * 7 + j]) : "v"(a), "v"(b));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 28; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int THREADS>
static float run(float* d, int blocks, int iters) {
    const int threads = THREADS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(threads), 0, 0, d, 16, 1.0001f, 0.5f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 1024 * sizeof(float));
    const int iters = 200000;
    for (int threads : {256, 512, 1024}) {
        float m, v, mv;
        if (threads == 256) { m = run<1, 256>(d, 256, iters); v = run<2, 256>(d, 256, iters); mv = run<3, 256>(d, 256, iters); }
        else if (threads == 512) { m = run<1, 512>(d, 256, iters); v = run<2, 512>(d, 256, iters); mv = run<3, 512>(d, 256, iters); }
        else { m = run<1, 1024>(d, 256, iters); v = run<2, 1024>(d, 256, iters); mv = run<3, 1024>(d, 256, iters); }
        double per = 1e6 / (double)(iters * 4);       // ns per (1 MFMA + 7 VALU) group
        printf("%d waves/SIMD (times are for ALL waves; per-wave instruction counts equal): MFMA only %.2f ms (%.1f ns/MFMA/wave), VALU only %.2f ms (%.2f ns per v_fma/wave), interleaved %.2f ms  -> sum %.2f, max %.2f\n",
               threads / 256, m, m * per, v, v * per / 7.0, mv, m + v, m > v ? m : v);
    }
    return 0;
}
