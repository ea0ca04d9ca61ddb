// gpe_fused.h -- fused jet-MLP kernels for gfx950 (CDNA4), hidden width H in {32, 64}.
//
// One wavefront (64 lanes) owns a tile of 16 collocation points and carries ALL C = 1+2d derivative channels
// of ALL H features of that tile through the whole network in registers; the H x H layers run on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 == fmaf chain).  No LDS and no barrier in the forward pass.
//
// Register/lane layout ("point-on-lane"): lane = m + 16 q  (m = point in tile 0..15, q = 0..3);
// for feature tile nt (16 features) the lane holds features n = 16 nt + 4 q + r, r = 0..3 (one f32x4).
// That is exactly the C/D layout of v_mfma_f32_16x16x4_f32 (col = lane&15, row = 4*(lane>>4)+r), and it is
// also a valid B-operand layout for the NEXT layer's MFMA once the k index inside a 16-feature tile is
// permuted to k = 16 kt + 4 q + s  (s = MFMA step 0..3): the A operand (the weights) is loaded from a copy
// pre-packed in that same permuted order, so layers chain with no data movement at all.
//
//   forward   Z^T[n][m]  = sum_k W[n][k]  A^T[k][m]      A-op = W  frag, B-op = activation regs
//   backward  Xb^T[k][m] = sum_n W[n][k]  Zb^T[n][m]     A-op = W^T frag, B-op = adjoint regs
//   weights   dW[n][k]   = sum_{c,m} Zb[n][m] X[k][m]    needs feature-on-lane operands: 16x16 tiles are
//                                                        transposed through a wave-private LDS scratch
// Gradients are accumulated per workgroup in LDS (ds_add_f32) over all tiles the workgroup processes and
// written once as a slab; k_grad_reduce sums the slabs in fixed order.
//
// Stored for the reverse pass (hidden layers 1..L-1; layer 0 is recomputed from x): per tile, per layer,
// per channel, per feature tile one f32x4 per lane -- "fragment native", 1 KiB per wave store, coalesced.
//   channel 0: t = tanh(z); channels 1..D: dz/dx_j; channels D+1..2D: d2z/dx_j^2
//
// Replaces the op sequences K1-K3, K12 of SURVEY 2.3 (refine/harmonic_pinn_simulation.py:121-125,158-172,358).
#pragma once
#include "gpe_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef GPE_FWD_WAVES
#define GPE_FWD_WAVES 2      // waves per SIMD the forward kernel is compiled for (C <= 5)
#endif
#ifndef GPE_BWD_WAVES
#define GPE_BWD_WAVES 2
#endif

#define F_PITCH 20   // floats per row of a transposition tile (16 + 4 pad; rows stay 16-B aligned)

GPE_DEV void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the 16 lanes of a DPP row (lanes with equal lane>>4); result valid in every lane of the row.
// Four v_add_f32 with DPP operands (quad_perm xor1, xor2, row_half_mirror, row_mirror): no LDS, no waitcnt.
template <int CTRL>
GPE_DEV float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
GPE_DEV float row_sum16(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return v;
}

// tile held point-on-lane (lane (m,q) reg r <-> row 4q+r, col m)  ->  feature-on-lane
// (lane (i,q') element s <-> row i, col 4q'+s)
// C tiles at once through C wave-private LDS tiles: one fence pair per batch instead of per tile.
#define F_TILE (16 * F_PITCH)
template <int C>
GPE_DEV void tiles_transpose(const f32x4 (&v)[C], f32x4 (&o)[C], float* T, int m, int q) {
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[c * F_TILE + (4 * q + r) * F_PITCH + m] = v[c][r];
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = *reinterpret_cast<const f32x4*>(&T[c * F_TILE + m * F_PITCH + 4 * q]);
}

// pack hidden-hidden weights (linear maps 1..L-1) in MFMA fragment order.
//   Wpk [j-1][nt][kt][lane][s] = W_j[16nt + (lane&15)][16kt + 4(lane>>4) + s]      (forward A operand)
//   WpkT[j-1][kt][nt][lane][s] = W_j[16nt + 4(lane>>4) + s][16kt + (lane&15)]      (backward A operand)
__global__ void k_pack_weights(NetDesc nd, int H, const float* __restrict__ theta, float* __restrict__ Wpk,
                               float* __restrict__ WpkT) {
    const int NT = H / 16;
    const int L = nd.n_lin - 1;
    int idx = blockIdx.x * 256 + threadIdx.x;
    int per = H * H;
    if (idx >= (L - 1) * per) return;
    int j = idx / per + 1, e = idx % per;
    int s = e & 3, lane = (e >> 2) & 63, t2 = e >> 8;   // t2 = nt*NT+kt (or kt*NT+nt)
    int a = t2 / NT, b = t2 % NT;
    int i = lane & 15, q = lane >> 4;
    const float* W = theta + nd.offW[j];
    Wpk[idx] = W[(16 * a + i) * H + 16 * b + 4 * q + s];           // a = nt, b = kt
    WpkT[idx] = W[(16 * b + 4 * q + s) * H + 16 * a + i];          // a = kt, b = nt
}

template <int H, int C, int NOUT>
__global__ __launch_bounds__(256, (C <= 5 ? GPE_FWD_WAVES : 1)) void f_forward(NetDesc nd, const float* __restrict__ theta,
                                                                 const float* __restrict__ Wpk,
                                                                 const float* __restrict__ x, float* __restrict__ stored,
                                                                 float* __restrict__ O, int64_t N, int64_t ld,
                                                                 int store_acts) {
    constexpr int D = (C - 1) / 2, NT = H / 16, NF = NT * 4;
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    const int L = nd.n_lin - 1;
    const int dim = nd.dim;
    const int64_t ntiles = (N + 15) >> 4;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float* W0 = theta + nd.offW[0];
    const float* b0 = theta + nd.offB[0];
    const float shift = nd.shift;

    for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
        const int64_t pm = tile * 16 + m;
        const bool valid = pm < N;
        const int64_t pl = valid ? pm : N - 1;
        float xv[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k < dim) xv[k] = x[pl * dim + k];

        float a_in[C][NF];
        // ---- layer 0 (K = dim <= 3): VALU -----------------------------------------------------------
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * nt + 4 * q + r;
                float z = b0[n];
                float wk[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 3; ++k) if (k < dim) { wk[k] = W0[n * dim + k]; z = fmaf(wk[k], xv[k], z); }
                float t = gpe_tanh(z);
                float s = fmaf(-t, t, 1.0f);
                a_in[0][nt * 4 + r] = t + shift;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    a_in[1 + j][nt * 4 + r] = s * wk[j];
                    a_in[1 + D + j][nt * 4 + r] = -2.0f * t * s * wk[j] * wk[j];
                }
            }
        }
        // ---- hidden -> hidden layers on the matrix cores -------------------------------------------------
        for (int j = 1; j < L; ++j) {
            const float* Wp = Wpk + (size_t)(j - 1) * H * H;
            const float* bj = theta + nd.offB[j];
            float a_out[C][NF];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 acc[C];
                acc[0] = *reinterpret_cast<const f32x4*>(&bj[16 * nt + 4 * q]);
#pragma unroll
                for (int c = 1; c < C; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(&Wp[((nt * NT + kt) * 64 + lane) * 4]);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], a_in[c][kt * 4 + s], acc[c], 0, 0, 0);
                }
                f32x4 tt;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = gpe_tanh(acc[0][r]);
                    tt[r] = t;
                    float s = fmaf(-t, t, 1.0f);
                    float ts2 = 2.0f * t * s;
                    a_out[0][nt * 4 + r] = t + shift;
#pragma unroll
                    for (int jd = 0; jd < D; ++jd) {
                        float zk = acc[1 + jd][r], zkk = acc[1 + D + jd][r];
                        a_out[1 + jd][nt * 4 + r] = s * zk;
                        a_out[1 + D + jd][nt * 4 + r] = fmaf(s, zkk, -ts2 * zk * zk);
                    }
                }
                if (store_acts) {
                    float* sp = stored + ((((size_t)tile * (L - 1) + (j - 1)) * C) * NT + nt) * 256 + lane * 4;
                    *reinterpret_cast<f32x4*>(sp) = tt;
#pragma unroll
                    for (int c = 1; c < C; ++c) *reinterpret_cast<f32x4*>(sp + (size_t)c * NT * 256) = acc[c];
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int f = 0; f < NF; ++f) a_in[c][f] = a_out[c][f];
        }
        // ---- output layer (n_out <= 2): VALU dot + reduction over the 4 q-lanes of a point ------------------
        const float* Wo = theta + nd.offW[L];
        const float* bo = theta + nd.offB[L];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float part[C];
#pragma unroll
            for (int c = 0; c < C; ++c) part[c] = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(&Wo[o * H + 16 * nt + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < C; ++c) part[c] = fmaf(w[r], a_in[c][nt * 4 + r], part[c]);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float v = part[c];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (c == 0) v += bo[o];
                if (q == 0 && valid) O[((int64_t)c * NOUT + o) * ld + pm] = v;
            }
        }
    }
}

// Reverse pass.  Ob = dLoss/dO ([C][NOUT][ld]).  gslab: [gridDim.x][Ppad] per-workgroup gradient slabs.
// Dynamic LDS: Ppad floats of gradient accumulators + 4 waves x 2 x 16 x F_PITCH floats transposition scratch.
template <int H, int C, int NOUT>
__global__ __launch_bounds__(256, (C <= 5 ? GPE_BWD_WAVES : 1)) void f_backward(NetDesc nd, const float* __restrict__ theta,
                                                                  const float* __restrict__ WpkT,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ stored,
                                                                  const float* __restrict__ Ob, float* __restrict__ gslab,
                                                                  int64_t N, int64_t ld, int Ppad) {
    constexpr int D = (C - 1) / 2, NT = H / 16, NF = NT * 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* gacc = lds;
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    const int wib = threadIdx.x >> 6;
    float* TT = lds + Ppad + wib * (C * F_TILE);      // C transposition tiles, private to this wave
    const int L = nd.n_lin - 1;
    const int dim = nd.dim;
    const float shift = nd.shift;
    const int64_t ntiles = (N + 15) >> 4;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float* W0 = theta + nd.offW[0];
    const float* b0 = theta + nd.offB[0];
    const float* Wo = theta + nd.offW[L];

    for (int i = threadIdx.x; i < Ppad; i += 256) gacc[i] = 0.f;
    __syncthreads();

    for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
        const int64_t pm = tile * 16 + m;
        const bool valid = pm < N;
        const int64_t pl = valid ? pm : N - 1;
        float xv[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k < dim) xv[k] = x[pl * dim + k];
        float ob[NOUT][C];
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int c = 0; c < C; ++c) ob[o][c] = valid ? Ob[((int64_t)c * NOUT + o) * ld + pm] : 0.f;

        // ---- output layer: dWout, dbout, adjoint into the last hidden layer, activation adjoint --------
        float zb[C][NF];
        {
            const float* sp0 = stored + (((size_t)tile * (L - 1) + (L - 2)) * C) * NT * 256 + lane * 4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 st[C];
#pragma unroll
                for (int c = 0; c < C; ++c) st[c] = *reinterpret_cast<const f32x4*>(sp0 + ((size_t)c * NT + nt) * 256);
                f32x4 wo[NOUT];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) wo[o] = *reinterpret_cast<const f32x4*>(&Wo[o * H + 16 * nt + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float zk[D > 0 ? D : 1], zkk[D > 0 ? D : 1], a[C], ab[C], zv[C];
#pragma unroll
                    for (int jd = 0; jd < D; ++jd) { zk[jd] = st[1 + jd][r]; zkk[jd] = st[1 + D + jd][r]; }
                    act_from_stored<D>(st[0][r], zk, zkk, shift, a);
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) {
                        float g = 0.f;
#pragma unroll
                        for (int c = 0; c < C; ++c) g = fmaf(ob[o][c], a[c], g);
                        g = row_sum16(g);
                        if (m == 0) atomicAdd(&gacc[nd.offW[L] + o * H + 16 * nt + 4 * q + r], g);
                    }
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        float v = 0.f;
#pragma unroll
                        for (int o = 0; o < NOUT; ++o) v = fmaf(wo[o][r], ob[o][c], v);
                        ab[c] = v;
                    }
                    act_adjoint<D>(st[0][r], zk, zkk, ab, zv);
#pragma unroll
                    for (int c = 0; c < C; ++c) zb[c][nt * 4 + r] = zv[c];
                }
            }
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                float g = row_sum16(ob[o][0]);
                if (lane == 0) atomicAdd(&gacc[nd.offB[L] + o], g);
            }
        }
        // ---- hidden -> hidden linear maps j = L-1 .. 1 ---------------------------------------------------
        for (int j = L - 1; j >= 1; --j) {
            // bias gradient of map j
            {
                float gb[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) gb[f] = row_sum16(zb[0][f]);
                if (m == 0) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) atomicAdd(&gacc[nd.offB[j] + 16 * (f >> 2) + 4 * q + (f & 3)], gb[f]);
                }
            }
            // B1: adjoint of the input jets  Xb^T = W_j^T Zb^T
            const float* WT = WpkT + (size_t)(j - 1) * H * H;
            float xb[C][NF];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x4 acc[C];
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(&WT[((kt * NT + nt) * 64 + lane) * 4]);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], zb[c][nt * 4 + s], acc[c], 0, 0, 0);
                }
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xb[c][kt * 4 + r] = acc[c][r];
            }
            // B2 + activation adjoint of hidden layer j-1, one input feature tile kt at a time
            const float* sp0 = stored + (((size_t)tile * (L - 1) + (j >= 2 ? j - 2 : 0)) * C) * NT * 256 + lane * 4;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x4 st[C];
                if (j >= 2) {
#pragma unroll
                    for (int c = 0; c < C; ++c) st[c] = *reinterpret_cast<const f32x4*>(sp0 + ((size_t)c * NT + kt) * 256);
                } else {   // hidden layer 0: recompute from x
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = 16 * kt + 4 * q + r;
                        float z = b0[n];
                        float wk[3] = {0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < 3; ++k) if (k < dim) { wk[k] = W0[n * dim + k]; z = fmaf(wk[k], xv[k], z); }
                        st[0][r] = gpe_tanh(z);
#pragma unroll
                        for (int jd = 0; jd < D; ++jd) { st[1 + jd][r] = wk[jd]; st[1 + D + jd][r] = 0.f; }
                    }
                }
                f32x4 xa[C];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float zk[D > 0 ? D : 1], zkk[D > 0 ? D : 1], a[C];
#pragma unroll
                    for (int jd = 0; jd < D; ++jd) { zk[jd] = st[1 + jd][r]; zkk[jd] = st[1 + D + jd][r]; }
                    act_from_stored<D>(st[0][r], zk, zkk, shift, a);
#pragma unroll
                    for (int c = 0; c < C; ++c) xa[c][r] = a[c];
                }
                f32x4 xt[C];
                tiles_transpose<C>(xa, xt, TT, m, q);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4 zv[C], zt[C];
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) zv[c][r] = zb[c][nt * 4 + r];
                    tiles_transpose<C>(zv, zt, TT, m, q);
                    f32x4 dw = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            dw = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[c][s], xt[c][s], dw, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        atomicAdd(&gacc[nd.offW[j] + (16 * nt + 4 * q + r) * H + 16 * kt + m], dw[r]);
                }
                // activation adjoint of hidden layer j-1 for this feature tile (overwrites xb in place)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float zk[D > 0 ? D : 1], zkk[D > 0 ? D : 1], ab[C], zv[C];
#pragma unroll
                    for (int jd = 0; jd < D; ++jd) { zk[jd] = st[1 + jd][r]; zkk[jd] = st[1 + D + jd][r]; }
#pragma unroll
                    for (int c = 0; c < C; ++c) ab[c] = xb[c][kt * 4 + r];
                    act_adjoint<D>(st[0][r], zk, zkk, ab, zv);
#pragma unroll
                    for (int c = 0; c < C; ++c) xb[c][kt * 4 + r] = zv[c];
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int f = 0; f < NF; ++f) zb[c][f] = xb[c][f];
        }
        // ---- linear map 0: z = W0 x + b0, dz/dx_k = W0[:,k] ------------------------------------------------
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int n = 16 * (f >> 2) + 4 * q + (f & 3);
            float g = row_sum16(zb[0][f]);
            if (m == 0) atomicAdd(&gacc[nd.offB[0] + n], g);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < dim) {
                    float v = zb[0][f] * xv[k];
                    if constexpr (C > 1) { if (k < D) v += zb[(1 + k) < C ? (1 + k) : 0][f]; }
                    v = row_sum16(v);
                    if (m == 0) atomicAdd(&gacc[nd.offW[0] + n * dim + k], v);
                }
            }
        }
    }
    __syncthreads();
    float* slab = gslab + (size_t)blockIdx.x * Ppad;
    for (int i = threadIdx.x; i < Ppad; i += 256) slab[i] = gacc[i];
}

// grad[i] += sum_b gslab[b][i].  Block = 64 parameters x 16 slab groups (1024 threads); fixed summation order, so the
// slab sum is deterministic for a given grid.
__global__ __launch_bounds__(1024) void k_grad_reduce(const float* __restrict__ gslab, int nslab, int Ppad, int P,
                                                       float* __restrict__ grad) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (i < P)
        for (int b = g; b < nslab; b += 16) s += gslab[(size_t)b * Ppad + i];
    red[g][lane] = s;
    __syncthreads();
    if (g == 0 && i < P) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        grad[i] += t;
    }
}
