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
                                                   float* __restrict__ Out, int64_t N, int64_t ld) {
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
            act_from_stored<D, E>(t, zk, zkk, nd.shift, a);
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
                act_from_stored<D, E>(t, zk, zkk, nd.shift, a);
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
