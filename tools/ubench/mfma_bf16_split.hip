// fp32-accurate GEMM from bf16 matrix instructions on gfx950?  (Round-3 planning data, not part of the engine.)
//   part 1 (rate): per K = 32 slab of a 16x16 tile, 8 x v_mfma_f32_16x16x4_f32 against 6 (or 3) x v_mfma_f32_16x16x32_bf16, alone and
//                  with VALU instructions interleaved (does the VALU co-execute with the bf16 matrix pipe? it does not with the fp32 one)
//   part 2 (accuracy): D = A B (16 x 64 by 64 x 16, entries uniform in [-1, 1]) by fp32 MFMA, by the 3-product split (hi hi, hi lo, lo hi:
//                  two bf16 pieces per operand) and by the 6-product split (three pieces: hh, hm, mh, hl, lh, mm), against fp64.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_bf16_split.hip -o build/ubench_mfma_bf16_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE bit 0: fp32 MFMAs (8 per slab), bit 1: bf16 MFMAs (NB per slab), bit 2: NV v_fma per slab interleaved
template <int MODE, int NB, int NV, int THREADS>
__global__ __launch_bounds__(THREADS) void rate(float* out, int iters, float a, float b) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 av = {a, a, a, a}, bv = {b, b, b, b};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {                               // four independent accumulator chains, one slab each
            if (MODE & 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
            }
            if (MODE & 2) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(u + j) & 3]) : "v"(av), "v"(bv));
                    if (MODE & 4) {
#pragma unroll
                        for (int t = 0; t < NV / NB; ++t)
                            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(j * (NV / NB) + t) & 15]) : "v"(a), "v"(b));
                    }
                }
            } else if (MODE & 4) {
#pragma unroll
                for (int t = 0; t < NV; ++t) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[t & 15]) : "v"(a), "v"(b));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NB, int NV, int THREADS>
static float run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<MODE, NB, NV, THREADS>), dim3(256), dim3(THREADS), 0, 0, d, 16, 1.0001f, 0.5f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate<MODE, NB, NV, THREADS>), dim3(256), dim3(THREADS), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

// ---- accuracy ----------------------------------------------------------------------------------------------------------------------
__device__ inline unsigned top16(float x) { return __builtin_bit_cast(unsigned, x) >> 16; }
__device__ inline float trunc16(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u); }
// eight fp32 values -> three bf16x8 pieces by truncation: x = h + m + l + O(2^-24 |x|), every piece exact in bf16
__device__ inline void split3(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    unsigned hh[8], mm[8], ll[8];
    for (int e = 0; e < 8; ++e) {
        const float hf = trunc16(x[e]);
        const float r1 = x[e] - hf;
        const float mf = trunc16(r1);
        const float r2 = r1 - mf;
        hh[e] = top16(hf); mm[e] = top16(mf); ll[e] = top16(r2);
    }
    u32x4 ph, pm, pl;
    for (int j = 0; j < 4; ++j) {
        ph[j] = hh[2 * j] | (hh[2 * j + 1] << 16);
        pm[j] = mm[2 * j] | (mm[2 * j + 1] << 16);
        pl[j] = ll[2 * j] | (ll[2 * j + 1] << 16);
    }
    h = __builtin_bit_cast(bf16x8, ph); m = __builtin_bit_cast(bf16x8, pm); l = __builtin_bit_cast(bf16x8, pl);
}

// A: [16][K] row-major, B: [K][16] row-major, D: [3][16][16] (fp32 MFMA, 3-product split, 6-product split); one wave
__global__ void accuracy(const float* A, const float* B, float* D, int K) {
    const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
    f32x4 d0 = {0, 0, 0, 0}, d3 = {0, 0, 0, 0}, d6 = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 4)                              // fp32: lane (i, kq) holds A[i][k0 + kq], B[k0 + kq][i]
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k0 + kq], B[(k0 + kq) * 16 + i], d0, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 32) {                            // bf16: lane (i, kq) holds A[i][k0 + 8kq .. +7], B[k0 + 8kq .. +7][i]
        float xa[8], xb[8];
        for (int e = 0; e < 8; ++e) { xa[e] = A[i * K + k0 + 8 * kq + e]; xb[e] = B[(k0 + 8 * kq + e) * 16 + i]; }
        bf16x8 ah, am, al, bh, bm, bl;
        split3(xa, ah, am, al);
        split3(xb, bh, bm, bl);
        // smallest terms first
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, d6, 0, 0, 0);
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d6, 0, 0, 0);
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d6, 0, 0, 0);
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, d6, 0, 0, 0);
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, d6, 0, 0, 0);
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d6, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, d3, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, d3, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d3, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) {                                   // lane (n = i, q = kq) reg r <-> D[4q + r][n]
        D[0 * 256 + (4 * kq + r) * 16 + i] = d0[r];
        D[1 * 256 + (4 * kq + r) * 16 + i] = d3[r];
        D[2 * 256 + (4 * kq + r) * 16 + i] = d6[r];
    }
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 1024 * sizeof(float));
    const int iters = 40000;
    printf("part 1: time for 256 workgroups x W waves/SIMD x %d iterations x 4 slabs (K = 32 of a 16x16 tile each)\n", iters);
#define ROW(T, W)                                                                                                                       \
    {                                                                                                                                   \
        const float f = run<1, 0, 0, T>(d, iters), b6 = run<2, 6, 0, T>(d, iters), b3 = run<2, 3, 0, T>(d, iters);                      \
        const float v24 = run<4, 6, 24, T>(d, iters), b6v = run<6, 6, 24, T>(d, iters), b6v48 = run<6, 6, 48, T>(d, iters);             \
        const float v48 = run<4, 6, 48, T>(d, iters);                                                                                   \
        printf("  %d wave(s)/SIMD: fp32 x8 %.2f ms | bf16 x6 %.2f ms (%.2fx) | bf16 x3 %.2f ms (%.2fx) | 24 v_fma alone %.2f ms, with bf16 x6 %.2f ms "  \
               "| 48 v_fma alone %.2f ms, with bf16 x6 %.2f ms\n", W, f, b6, f / b6, b3, f / b3, v24, b6v, v48, b6v48);                 \
    }
    ROW(256, 1) ROW(512, 2) ROW(1024, 4)
    // ---- accuracy ----
    const int K = 64;
    std::vector<float> A(16 * K), B(K * 16);
    srand(1);
    for (auto& x : A) x = 2.f * rand() / RAND_MAX - 1.f;
    for (auto& x : B) x = 2.f * rand() / RAND_MAX - 1.f;
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 3 * 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    std::vector<float> D(3 * 256);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double err[3] = {0, 0, 0}, ref_max = 0;
    for (int i = 0; i < 16; ++i)
        for (int n = 0; n < 16; ++n) {
            double r = 0;
            for (int k = 0; k < K; ++k) r += (double)A[i * K + k] * (double)B[k * 16 + n];
            ref_max = fmax(ref_max, fabs(r));
            for (int v = 0; v < 3; ++v) err[v] = fmax(err[v], fabs((double)D[v * 256 + i * 16 + n] - r));
        }
    printf("part 2: K = %d, max |D| = %.3f; max abs error vs fp64: fp32 MFMA %.3e | bf16 x3 %.3e | bf16 x6 %.3e\n", K, ref_max, err[0], err[1], err[2]);
    return 0;
}
