// gpe_generic.h -- generic (any width <= 1024, dim <= 3, n_out <= 2) layer-materialised jet-MLP kernels.
// VALU only, one thread per collocation point, feature-major buffers [C][width][ld] (point index contiguous,
// so every global access is coalesced across the wave).  This set is the coverage / cross-check path and the
// fallback for shapes the fused MFMA set (gpe_fused.h) does not take.
//
// Stored per hidden layer h (for the reverse pass):  S_h[0] = t = tanh(z),  S_h[1+j] = dz/dx_j,
// S_h[1+D+e] = second-order channel e (d2z/dx_e^2, or the Laplacian when E = 1 < D; gpe_common.h).  Activation jets are recomputed from S on load (gpe_common.h:act_from_stored).
// Replaces: nn.Sequential forward + the two torch.autograd.grad(create_graph=True) calls + loss.backward() of
// refine/harmonic_pinn_simulation.py:121-125,158-172,358 (2D: src/gross_pitaevskii_2D.py:183-188).
#pragma once
#include "gpe_common.h"

#define G_FB 4   // output features per thread (narrow layers)
#define G_FBW 16 // output features per thread for layers at least 64 wide: the activation recompute and the input loads of a
                 // point are shared by 16 outputs instead of 4 (cfg5: the layer kernels were recompute- and load-bound)

// lin: index of the linear map.  Sprev: stored of hidden layer lin-1 (NULL for lin==0).  Out: stored of hidden
// layer lin, or the output jets O when lin == n_lin-1.
template <int C, int E, int FB>
__global__ __launch_bounds__(256) void g_fwd_layer(NetDesc nd, int lin, const float* __restrict__ theta,
                                                   Pts x, const float* __restrict__ Sprev,
                                                   float* __restrict__ Out, int64_t N, int64_t ld,
                                                   const float* __restrict__ Sskip /* stored of hidden layer nd.skip[lin], or NULL */) {
    constexpr int D = C - 1 - E;
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= N) return;
    const int K = nd.width[lin], Ho = nd.width[lin + 1];
    const int n0 = blockIdx.y * FB;
    const float* W = theta + nd.offW[lin];
    const float* b = theta + nd.offB[lin];
    float acc[FB][C];
#pragma unroll
    for (int f = 0; f < FB; ++f) {
        int n = min(n0 + f, Ho - 1);
        acc[f][0] = b[n];
#pragma unroll
        for (int c = 1; c < C; ++c) acc[f][c] = 0.f;
    }
    if (lin == 0) {
        for (int k = 0; k < K; ++k) {
            float xk = pts_at(x, m, K, k);
#pragma unroll
            for (int f = 0; f < FB; ++f) {
                int n = min(n0 + f, Ho - 1);
                float w = W[n * K + k];
                acc[f][0] = fmaf(w, xk, acc[f][0]);
                if (C > 1) {
#pragma unroll
                    for (int j = 0; j < D; ++j) if (j == k) acc[f][1 + j] = w;
                }
            }
        }
    } else {
        for (int k = 0; k < K; ++k) {
            float t = Sprev[((int64_t)0 * K + k) * ld + m];
            float zk[D > 0 ? D : 1], zkk[E > 0 ? E : 1], a[C];
#pragma unroll
            for (int j = 0; j < D; ++j) zk[j] = Sprev[((int64_t)(1 + j) * K + k) * ld + m];
#pragma unroll
            for (int j = 0; j < E; ++j) zkk[j] = Sprev[((int64_t)(1 + D + j) * K + k) * ld + m];
            act_from_stored<D, E>(t, zk, zkk, nd.shiftv[lin - 1], a);
#pragma unroll
            for (int f = 0; f < FB; ++f) {
                int n = min(n0 + f, Ho - 1);
                float w = W[n * K + k];
#pragma unroll
                for (int c = 0; c < C; ++c) acc[f][c] = fmaf(w, a[c], acc[f][c]);
            }
        }
    }
    const bool last = (lin == nd.n_lin - 1);
    if (Sskip) {      // residual block (refine/box_to_gaussian_pinn_simulation.py:58-62): z += activation jets of the block's input
        const float sh = nd.shiftv[nd.skip[lin]];
#pragma unroll
        for (int f = 0; f < FB; ++f) {
            const int n = min(n0 + f, Ho - 1);
            float zk[D > 0 ? D : 1], zkk[E > 0 ? E : 1], a[C];
#pragma unroll
            for (int j = 0; j < D; ++j) zk[j] = Sskip[((int64_t)(1 + j) * Ho + n) * ld + m];
#pragma unroll
            for (int j = 0; j < E; ++j) zkk[j] = Sskip[((int64_t)(1 + D + j) * Ho + n) * ld + m];
            act_from_stored<D, E>(Sskip[((int64_t)0 * Ho + n) * ld + m], zk, zkk, sh, a);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[f][c] += a[c];
        }
    }
#pragma unroll
    for (int f = 0; f < FB; ++f) {
        int n = n0 + f;
        if (n >= Ho) break;
        if (!last) acc[f][0] = gpe_tanh(acc[f][0]);
#pragma unroll
        for (int c = 0; c < C; ++c) Out[((int64_t)c * Ho + n) * ld + m] = acc[f][c];
    }
}

// Zb = act_adjoint(Ab, S_h) in place.  grid (ceil(N/256), H).
template <int C, int E>
__global__ __launch_bounds__(256) void g_bwd_act(int H, const float* __restrict__ S, float* __restrict__ A,
                                                 int64_t N, int64_t ld) {
    constexpr int D = C - 1 - E;
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= N) return;
    int n = blockIdx.y;
    float t = S[((int64_t)0 * H + n) * ld + m];
    float zk[D > 0 ? D : 1], zkk[E > 0 ? E : 1], ab[C], zb[C];
#pragma unroll
    for (int j = 0; j < D; ++j) zk[j] = S[((int64_t)(1 + j) * H + n) * ld + m];
#pragma unroll
    for (int j = 0; j < E; ++j) zkk[j] = S[((int64_t)(1 + D + j) * H + n) * ld + m];
#pragma unroll
    for (int c = 0; c < C; ++c) ab[c] = A[((int64_t)c * H + n) * ld + m];
    act_adjoint<D, E>(t, zk, zkk, ab, zb);
#pragma unroll
    for (int c = 0; c < C; ++c) A[((int64_t)c * H + n) * ld + m] = zb[c];
}

// A[i] += B[i]: the adjoint arriving over a skip connection joins the adjoint of the skipped-from activations
__global__ void g_add(float* __restrict__ A, const float* __restrict__ B, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) A[i] += B[i];
}

// Aprev[c][k][m] = sum_n W[n][k] Zb[c][n][m].   grid (ceil(N/256), ceil(K/G_FB)).
template <int C, int FB>
__global__ __launch_bounds__(256) void g_bwd_data(NetDesc nd, int lin, const float* __restrict__ theta,
                                                  const float* __restrict__ Zb, float* __restrict__ Aprev, int64_t N,
                                                  int64_t ld) {
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= N) return;
    const int K = nd.width[lin], Ho = nd.width[lin + 1];
    const int k0 = blockIdx.y * FB;
    const float* W = theta + nd.offW[lin];
    float acc[FB][C];
#pragma unroll
    for (int f = 0; f < FB; ++f)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[f][c] = 0.f;
    for (int n = 0; n < Ho; ++n) {
        float z[C];
#pragma unroll
        for (int c = 0; c < C; ++c) z[c] = Zb[((int64_t)c * Ho + n) * ld + m];
#pragma unroll
        for (int f = 0; f < FB; ++f) {
            int k = min(k0 + f, K - 1);
            float w = W[n * K + k];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[f][c] = fmaf(w, z[c], acc[f][c]);
        }
    }
#pragma unroll
    for (int f = 0; f < FB; ++f) {
        int k = k0 + f;
        if (k >= K) break;
#pragma unroll
        for (int c = 0; c < C; ++c) Aprev[((int64_t)c * K + k) * ld + m] = acc[f][c];
    }
}

// grad W[n][k] += sum_{c,m} Zb[c][n][m] * A[c][k][m],  grad b[n] += sum_m Zb[0][n][m].
// One block per (NB-row block of n, 16-wide k block); the block walks all points -> deterministic, no atomics.  The activation
// jets A of a point are recomputed once per k and shared by the NB rows (NB = 8 for wide layers: with one row per block the
// kernel recomputed them Ho times and was 64 % of a cfg5 step).
// Narrow outputs (Ho < 8, the output layer): KB = 1, one block per (n, k), so that the grid still has K blocks.
#define G_KB 16
template <int C, int E, int NB, int KB>
__global__ __launch_bounds__(256) void g_bwd_weight(NetDesc nd, int lin, Pts x,
                                                    const float* __restrict__ Sprev, const float* __restrict__ Zb,
                                                    float* __restrict__ grad, int64_t N, int64_t ld) {
    constexpr int D = C - 1 - E;
    __shared__ double red[4];
    const int K = nd.width[lin], Ho = nd.width[lin + 1];
    const int n0 = blockIdx.x * NB, k0 = blockIdx.y * KB;
    float p[NB][KB];
    float pb[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        pb[r] = 0.f;
#pragma unroll
        for (int i = 0; i < KB; ++i) p[r][i] = 0.f;
    }
    for (int64_t m = threadIdx.x; m < N; m += 256) {
        float z[NB][C];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const int n = min(n0 + r, Ho - 1);
#pragma unroll
            for (int c = 0; c < C; ++c) z[r][c] = Zb[((int64_t)c * Ho + n) * ld + m];
            pb[r] += z[r][0];
        }
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            int k = k0 + i;
            if (k >= K) break;
            if (lin == 0) {
                const float xk = pts_at(x, m, K, k);
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    float v = z[r][0] * xk;
                    if (C > 1) {
#pragma unroll
                        for (int j = 0; j < D; ++j) if (j == k) v += z[r][1 + j];
                    }
                    p[r][i] += v;
                }
            } else {
                float t = Sprev[((int64_t)0 * K + k) * ld + m];
                float zk[D > 0 ? D : 1], zkk[E > 0 ? E : 1], a[C];
#pragma unroll
                for (int j = 0; j < D; ++j) zk[j] = Sprev[((int64_t)(1 + j) * K + k) * ld + m];
#pragma unroll
                for (int j = 0; j < E; ++j) zkk[j] = Sprev[((int64_t)(1 + D + j) * K + k) * ld + m];
                act_from_stored<D, E>(t, zk, zkk, nd.shiftv[lin - 1], a);
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    float v = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) v = fmaf(z[r][c], a[c], v);
                    p[r][i] += v;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        const int n = n0 + r;
        if (n >= Ho) break;
        for (int i = 0; i < KB; ++i) {
            int k = k0 + i;
            if (k >= K) break;
            double s = block_sum_256((double)p[r][i], red);
            if (threadIdx.x == 0) grad[nd.offW[lin] + n * K + k] += (float)s;
        }
        if (blockIdx.y == 0) {
            double s = block_sum_256((double)pb[r], red);
            if (threadIdx.x == 0) grad[nd.offB[lin] + n] += (float)s;
        }
    }
}

// ---- weight gradient of a wide hidden->hidden map on the matrix cores ------------------------------------------------------
// dW[n][k] += sum_c sum_m Zb[c][n][m] A_c[k][m]  is a GEMM with the POINTS as the contraction index.  The generic set's buffers
// are feature-major with the point index contiguous, so both MFMA operands of v_mfma_f32_16x16x4_f32 come straight from global
// memory as float4 loads (lane (i, kq): row n0+i resp. k0+i, points m0+4kq..m0+4kq+3; element s of the float4 is the k-slot of
// the s-th of four MFMAs) -- no LDS, no transposes.  One wave owns a 64 x 64 block of dW (16 accumulator tiles) and a chunk of
// the points (split-K); the activation jets A are recomputed per wave from the stored (t, z_k, z_L).  Partial blocks are added to
// the gradient with float atomics (the one place where the generic set is not bitwise reproducible).
// grid (Ho/64, K/64, chunks/4), block 256 = 4 waves on consecutive chunks.  Needs Ho % 64 == 0, K % 64 == 0, lin >= 1.
typedef float g_f32x4 __attribute__((ext_vector_type(4)));
template <int C, int E>
__global__ __launch_bounds__(256, 1) void g_bwd_weight_mfma(NetDesc nd, int lin, const float* __restrict__ Sprev,
                                                           const float* __restrict__ Zb, float* __restrict__ grad, int64_t N,
                                                           int64_t ld, int64_t chunk) {
    constexpr int D = C - 1 - E;
    const int K = nd.width[lin], Ho = nd.width[lin + 1];
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4, w = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int64_t c0 = ((int64_t)blockIdx.z * 4 + w) * chunk;
    const int64_t c1 = min(c0 + chunk, N);
    if (c0 >= N) return;                              // no barriers in this kernel
    g_f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (g_f32x4){0.f, 0.f, 0.f, 0.f};
    float pb[4] = {0.f, 0.f, 0.f, 0.f};
    auto ld4 = [&](const float* base, int64_t m) -> g_f32x4 {          // 4 consecutive points, zero past the end (ld is a multiple of 64)
        g_f32x4 v = *reinterpret_cast<const g_f32x4*>(base + m);
        if (m + 3 >= c1) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) if (m + s2 >= c1) v[s2] = 0.f;
        }
        return v;
    };
    for (int64_t m0 = c0; m0 < c1; m0 += 16) {
        const int64_t m = m0 + 4 * kq;
        // activation jets of the 4 K tiles at this lane's 4 points
        g_f32x4 a[4][C];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int k = k0 + 16 * kt + i;
            g_f32x4 st[C];
#pragma unroll
            for (int c = 0; c < C; ++c) st[c] = *reinterpret_cast<const g_f32x4*>(Sprev + ((int64_t)c * K + k) * ld + m);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                float zk[D > 0 ? D : 1], zkk[E > 0 ? E : 1], av[C];
#pragma unroll
                for (int j = 0; j < D; ++j) zk[j] = st[1 + j][s2];
#pragma unroll
                for (int j = 0; j < E; ++j) zkk[j] = st[1 + D + j][s2];
                act_from_stored<D, E>(st[0][s2], zk, zkk, nd.shiftv[lin - 1], av);
#pragma unroll
                for (int c = 0; c < C; ++c) a[kt][c][s2] = av[c];
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            g_f32x4 za[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) za[nt] = ld4(Zb + ((int64_t)c * Ho + n0 + 16 * nt + i) * ld, m);
            if (c == 0) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) pb[nt] += (za[nt][0] + za[nt][1]) + (za[nt][2] + za[nt][3]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[nt][s2], a[kt][c][s2], acc[nt][kt], 0, 0, 0);
        }
    }
    // D layout: lane (col = i, q = kq), element r <-> row 4q + r
    float* gW = grad + nd.offW[lin];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                atomicAdd(&gW[(int64_t)(n0 + 16 * nt + 4 * kq + r) * K + k0 + 16 * kt + i], acc[nt][kt][r]);
    if (blockIdx.y == 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float v = pb[nt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) atomicAdd(&grad[nd.offB[lin] + n0 + 16 * nt + i], v);
        }
    }
}

// ---- wide layers on the matrix cores: forward map and adjoint map ----------------------------------------------------------
// Same idea as g_bwd_weight_mfma: with feature-major buffers (point index contiguous) the operands of v_mfma_f32_16x16x4_f32 are
// plain global loads.  One wave = one 16-point tile x 64 output features (4 accumulator tiles per channel).
//   forward:  Out[c][n][m] = sum_k W[n][k] A_c[k][m] (+ b[n] on the value channel), tanh on the value channel of hidden layers;
//             A = W rows (float4 along k), B = activation jets recomputed from the stored (t, z_k, z_L) of the previous layer
//   adjoint:  Aprev[c][k][m] = sum_n W[n][k] Zb[c][n][m];  A = W^T (scalar loads, lanes along k), B = Zb
// grid (ceil(N/16/4), Ho/64) resp. (ceil(N/16/4), K/64), block 256 = 4 waves on consecutive point tiles.
// Needs: both widths multiples of 64, lin >= 1.
template <int C, int E>
__global__ __launch_bounds__(256) void g_fwd_layer_mfma(NetDesc nd, int lin, const float* __restrict__ theta,
                                                        const float* __restrict__ Sprev, float* __restrict__ Out, int64_t N,
                                                        int64_t ld, const float* __restrict__ Sskip /* stored of hidden layer nd.skip[lin], or NULL */) {
    constexpr int D = C - 1 - E;
    const int K = nd.width[lin], Ho = nd.width[lin + 1];
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4, w = threadIdx.x >> 6;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + w) * 16;
    if (m0 >= N) return;
    const int n0 = blockIdx.y * 64;
    const float* W = theta + nd.offW[lin];
    const float* bias = theta + nd.offB[lin];
    const int64_t mp = m0 + i;                                   // this lane's point as B-operand column / D column
    g_f32x4 acc[4][C];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        acc[nt][0] = *reinterpret_cast<const g_f32x4*>(&bias[n0 + 16 * nt + 4 * kq]);
#pragma unroll
        for (int c = 1; c < C; ++c) acc[nt][c] = (g_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int k0 = 0; k0 < K; k0 += 16) {
        g_f32x4 wv[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wv[nt] = *reinterpret_cast<const g_f32x4*>(&W[(int64_t)(n0 + 16 * nt + i) * K + k0 + 4 * kq]);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int k = k0 + 4 * kq + s2;
            float zk[D > 0 ? D : 1], zkk[E > 0 ? E : 1], a[C];
            const float t = Sprev[((int64_t)0 * K + k) * ld + mp];
#pragma unroll
            for (int j = 0; j < D; ++j) zk[j] = Sprev[((int64_t)(1 + j) * K + k) * ld + mp];
#pragma unroll
            for (int j = 0; j < E; ++j) zkk[j] = Sprev[((int64_t)(1 + D + j) * K + k) * ld + mp];
            act_from_stored<D, E>(t, zk, zkk, nd.shiftv[lin - 1], a);
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][s2], a[c], acc[nt][c], 0, 0, 0);
        }
    }
    const bool last = (lin == nd.n_lin - 1);
    if (mp < N) {
        const float sh = Sskip ? nd.shiftv[nd.skip[lin]] : 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * nt + 4 * kq + r;
                float sk[C];
#pragma unroll
                for (int c = 0; c < C; ++c) sk[c] = 0.f;
                if (Sskip) {      // residual block (refine/box_to_gaussian_pinn_simulation.py:58-62): z += activation jets of the block's input
                    float zk[D > 0 ? D : 1], zkk[E > 0 ? E : 1];
#pragma unroll
                    for (int j = 0; j < D; ++j) zk[j] = Sskip[((int64_t)(1 + j) * Ho + n) * ld + mp];
#pragma unroll
                    for (int j = 0; j < E; ++j) zkk[j] = Sskip[((int64_t)(1 + D + j) * Ho + n) * ld + mp];
                    act_from_stored<D, E>(Sskip[((int64_t)0 * Ho + n) * ld + mp], zk, zkk, sh, sk);
                }
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float v = acc[nt][c][r] + sk[c];
                    if (c == 0 && !last) v = gpe_tanh(v);
                    Out[((int64_t)c * Ho + n) * ld + mp] = v;
                }
            }
    }
}

template <int C>
__global__ __launch_bounds__(256) void g_bwd_data_mfma(NetDesc nd, int lin, const float* __restrict__ theta,
                                                       const float* __restrict__ Zb, float* __restrict__ Aprev, int64_t N,
                                                       int64_t ld) {
    const int K = nd.width[lin], Ho = nd.width[lin + 1];
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4, w = threadIdx.x >> 6;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + w) * 16;
    if (m0 >= N) return;
    const int k0 = blockIdx.y * 64;
    const float* W = theta + nd.offW[lin];
    const int64_t mp = m0 + i;
    g_f32x4 acc[4][C];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[kt][c] = (g_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < Ho; n0 += 16) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int n = n0 + 4 * kq + s2;
            float wv[4], z[C];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) wv[kt] = W[(int64_t)n * K + k0 + 16 * kt + i];
#pragma unroll
            for (int c = 0; c < C; ++c) z[c] = Zb[((int64_t)c * Ho + n) * ld + mp];
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    acc[kt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[kt], z[c], acc[kt][c], 0, 0, 0);
        }
    }
    if (mp < N) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + 16 * kt + 4 * kq + r;
#pragma unroll
                for (int c = 0; c < C; ++c) Aprev[((int64_t)c * K + k) * ld + mp] = acc[kt][c][r];
            }
    }
}

// Block-cooperative version of g_bwd_weight_mfma: the 4 waves of a block own a 2 x 2 arrangement of 64 x 64 blocks of dW
// (128 x 128 per block) and walk the SAME point chunk; per 16-point step every wave loads / recomputes a quarter of the two operand
// panels (128 rows of Zb, the activation jets of 128 input features) into LDS, so the panels are fetched from L2 and recomputed once
// per block instead of once per wave: half the L2 traffic and a quarter of the recompute per MFMA.
// grid (Ho/128, K/128, chunks), block 256.  Needs Ho % 128 == 0, K % 128 == 0, lin >= 1.  LDS: 2 panels x C x 128 x 16 floats.
template <int C, int E>
__global__ __launch_bounds__(256, 1) void g_bwd_weight_mfma2(NetDesc nd, int lin, const float* __restrict__ Sprev,
                                                            const float* __restrict__ Zb, float* __restrict__ grad, int64_t N,
                                                            int64_t ld, int64_t chunk) {
    constexpr int D = C - 1 - E;
    extern __shared__ __attribute__((aligned(16))) float g_panels[];   // dynamic: 2 * C * 128 * 16 floats (80 KB for C = 5)
    float (*PZ)[128][16] = reinterpret_cast<float (*)[128][16]>(g_panels);                    // Zb panel:  [channel][row n][point]
    float (*PA)[128][16] = reinterpret_cast<float (*)[128][16]>(g_panels + C * 128 * 16);     // jets panel: [channel][col k][point]
    const int K = nd.width[lin], Ho = nd.width[lin + 1];
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4, w = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
    const int wn = (w >> 1) * 64, wk = (w & 1) * 64;                   // this wave's 64 x 64 block inside the 128 x 128 tile
    const int64_t c0 = (int64_t)blockIdx.z * chunk;
    const int64_t c1 = min(c0 + chunk, N);
    g_f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (g_f32x4){0.f, 0.f, 0.f, 0.f};
    float pb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t m0 = c0; m0 < c1; m0 += 16) {                         // c0 < N for every block of the grid
        const int64_t m = m0 + 4 * kq;
        __syncthreads();                                               // previous step's panel reads are done
        // this wave stages rows / columns [32w, 32w + 32) of both panels: lane (i, kq) handles rows 32w + i and 32w + 16 + i
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rr = 32 * w + 16 * h + i;
            g_f32x4 st[C];
#pragma unroll
            for (int c = 0; c < C; ++c) st[c] = *reinterpret_cast<const g_f32x4*>(Sprev + ((int64_t)c * K + k0 + rr) * ld + m);
            g_f32x4 av[C];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                float zk[D > 0 ? D : 1], zkk[E > 0 ? E : 1], a1[C];
#pragma unroll
                for (int j = 0; j < D; ++j) zk[j] = st[1 + j][s2];
#pragma unroll
                for (int j = 0; j < E; ++j) zkk[j] = st[1 + D + j][s2];
                act_from_stored<D, E>(st[0][s2], zk, zkk, nd.shiftv[lin - 1], a1);
#pragma unroll
                for (int c = 0; c < C; ++c) av[c][s2] = a1[c];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                *reinterpret_cast<g_f32x4*>(&PA[c][rr][4 * kq]) = av[c];
                g_f32x4 z = *reinterpret_cast<const g_f32x4*>(Zb + ((int64_t)c * Ho + n0 + rr) * ld + m);
                if (m + 3 >= c1) {
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) if (m + s2 >= c1) z[s2] = 0.f;
                }
                *reinterpret_cast<g_f32x4*>(&PZ[c][rr][4 * kq]) = z;
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < C; ++c) {
            g_f32x4 za[4], ab[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                za[t] = *reinterpret_cast<const g_f32x4*>(&PZ[c][wn + 16 * t + i][4 * kq]);
                ab[t] = *reinterpret_cast<const g_f32x4*>(&PA[c][wk + 16 * t + i][4 * kq]);
            }
            if (c == 0 && wk == 0) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) pb[nt] += (za[nt][0] + za[nt][1]) + (za[nt][2] + za[nt][3]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[nt][s2], ab[kt][s2], acc[nt][kt], 0, 0, 0);
        }
    }
    float* gW = grad + nd.offW[lin];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                atomicAdd(&gW[(int64_t)(n0 + wn + 16 * nt + 4 * kq + r) * K + k0 + wk + 16 * kt + i], acc[nt][kt][r]);
    if (blockIdx.y == 0 && wk == 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float v = pb[nt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) atomicAdd(&grad[nd.offB[lin] + n0 + wn + 16 * nt + i], v);
        }
    }
}

// Block-cooperative forward map: the 4 waves of a block take the SAME 16-point tile and 4 consecutive 64-feature output blocks;
// the activation jets of the tile (B operand) are recomputed once per block, 64 input features at a time, into an LDS panel
// [channel][point][64 k] (a lane reads its 4 consecutive k as one ds_read_b128).  grid (ceil(N/16), Ho/256), block 256.
// Needs Ho % 256 == 0, K % 64 == 0, lin >= 1.
template <int C, int E>
__global__ __launch_bounds__(256) void g_fwd_layer_mfma2(NetDesc nd, int lin, const float* __restrict__ theta,
                                                         const float* __restrict__ Sprev, float* __restrict__ Out, int64_t N,
                                                         int64_t ld) {
    constexpr int D = C - 1 - E;
    __shared__ __attribute__((aligned(16))) float PB[C][16][68];       // +4 pad: rows of different points start in different banks
    const int K = nd.width[lin], Ho = nd.width[lin + 1];
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4, w = threadIdx.x >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * 16;
    const int n0 = blockIdx.y * 256 + 64 * w;
    const float* W = theta + nd.offW[lin];
    const float* bias = theta + nd.offB[lin];
    const int64_t mp = m0 + i;
    g_f32x4 acc[4][C];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        acc[nt][0] = *reinterpret_cast<const g_f32x4*>(&bias[n0 + 16 * nt + 4 * kq]);
#pragma unroll
        for (int c = 1; c < C; ++c) acc[nt][c] = (g_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int kb = 0; kb < K; kb += 64) {
        __syncthreads();
        // wave w recomputes the jets of input features kb + 16w .. kb + 16w + 15 at the 16 points: lane (point i, kq) -> 4 features
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int kl = 16 * w + 4 * kq + s2, k = kb + kl;
            float zk[D > 0 ? D : 1], zkk[E > 0 ? E : 1], a[C];
            const float t = Sprev[((int64_t)0 * K + k) * ld + mp];
#pragma unroll
            for (int j = 0; j < D; ++j) zk[j] = Sprev[((int64_t)(1 + j) * K + k) * ld + mp];
#pragma unroll
            for (int j = 0; j < E; ++j) zkk[j] = Sprev[((int64_t)(1 + D + j) * K + k) * ld + mp];
            act_from_stored<D, E>(t, zk, zkk, nd.shiftv[lin - 1], a);
#pragma unroll
            for (int c = 0; c < C; ++c) PB[c][i][kl] = a[c];
        }
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < 64; k0 += 16) {
            g_f32x4 wv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                wv[nt] = *reinterpret_cast<const g_f32x4*>(&W[(int64_t)(n0 + 16 * nt + i) * K + kb + k0 + 4 * kq]);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const g_f32x4 bv = *reinterpret_cast<const g_f32x4*>(&PB[c][i][k0 + 4 * kq]);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[nt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][s2], bv[s2], acc[nt][c], 0, 0, 0);
            }
        }
    }
    const bool last = (lin == nd.n_lin - 1);
    if (mp < N) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * nt + 4 * kq + r;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float v = acc[nt][c][r];
                    if (c == 0 && !last) v = gpe_tanh(v);
                    Out[((int64_t)c * Ho + n) * ld + mp] = v;
                }
            }
    }
}

// Block-cooperative adjoint map: 4 waves = the same 16-point tile x 4 consecutive 64-feature blocks of the INPUT side; the adjoint
// jets Zb of the tile (B operand) are staged once per block, 64 output features at a time, in an LDS panel [channel][point][64 n].
// grid (ceil(N/16), K/256), block 256.  Needs K % 256 == 0, Ho % 64 == 0.
// Sact != NULL: the stored (t, z_k, z_L) of the layer the adjoint lands on -- the activation adjoint (g_bwd_act) is applied in the
// epilogue, all channels of a (feature, point) pair being in one lane, so that pass and its round trip through HBM disappear.
template <int C, int E>
__global__ __launch_bounds__(256) void g_bwd_data_mfma2(NetDesc nd, int lin, const float* __restrict__ theta,
                                                        const float* __restrict__ Zb, float* __restrict__ Aprev,
                                                        const float* __restrict__ Sact, int64_t N, int64_t ld) {
    constexpr int D = C - 1 - E;
    __shared__ __attribute__((aligned(16))) float PB[C][16][68];
    const int K = nd.width[lin], Ho = nd.width[lin + 1];
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4, w = threadIdx.x >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * 16;
    const int k0 = blockIdx.y * 256 + 64 * w;
    const float* W = theta + nd.offW[lin];
    const int64_t mp = m0 + i;
    g_f32x4 acc[4][C];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[kt][c] = (g_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int nb = 0; nb < Ho; nb += 64) {
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int nl = 16 * w + 4 * kq + s2;
#pragma unroll
            for (int c = 0; c < C; ++c) PB[c][i][nl] = Zb[((int64_t)c * Ho + nb + nl) * ld + mp];
        }
        __syncthreads();
#pragma unroll
        for (int n0 = 0; n0 < 64; n0 += 16) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int n = nb + n0 + 4 * kq + s2;
                float wv[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) wv[kt] = W[(int64_t)n * K + k0 + 16 * kt + i];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float z = PB[c][i][n0 + 4 * kq + s2];
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        acc[kt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[kt], z, acc[kt][c], 0, 0, 0);
                }
            }
        }
    }
    if (mp < N) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + 16 * kt + 4 * kq + r;
                float ab[C], zv[C];
#pragma unroll
                for (int c = 0; c < C; ++c) ab[c] = acc[kt][c][r];
                if (Sact) {
                    float zk[D > 0 ? D : 1], zkk[E > 0 ? E : 1];
                    const float t = Sact[((int64_t)0 * K + k) * ld + mp];
#pragma unroll
                    for (int j = 0; j < D; ++j) zk[j] = Sact[((int64_t)(1 + j) * K + k) * ld + mp];
#pragma unroll
                    for (int j = 0; j < E; ++j) zkk[j] = Sact[((int64_t)(1 + D + j) * K + k) * ld + mp];
                    act_adjoint<D, E>(t, zk, zkk, ab, zv);
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) zv[c] = ab[c];
                }
#pragma unroll
                for (int c = 0; c < C; ++c) Aprev[((int64_t)c * K + k) * ld + mp] = zv[c];
            }
    }
}
