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

#include "gpe_mfma_util.h"
#include "gpe_head.h"

#ifndef GPE_FWD_WAVES
#define GPE_FWD_WAVES 2      // waves per SIMD the forward kernel is compiled for (C <= 5)
#endif
#ifndef GPE_BWD_WAVES
#define GPE_BWD_WAVES 2
#endif
#ifndef GPE_COOP_PRIO
#ifndef GPE_FCOOP_ALT_PRIO
#define GPE_FCOOP_ALT_PRIO 0   // f_forward_coop<128>: SIMD partners alternate s_setprio per K tile (as w_forward does)
#endif
#define GPE_COOP_PRIO 1      // s_setprio level of the product phases of f_backward_coop (H <= 64); 0 switches it off
#endif
#ifndef GPE_FCOOP_SWP
#ifndef GPE_COOP_WREG_MAX
#define GPE_COOP_WREG_MAX 5  // f_backward_coop, H <= 64: most hidden->hidden maps whose W^T slices stay in registers (more: streamed from L2 as for H = 128)
#endif
#define GPE_FCOOP_SWP 0      // f_forward_coop<128>: B fragments of K tile kt+1 requested before the products of tile kt
#endif
#ifndef GPE_PIPE_G0REG
#define GPE_PIPE_G0REG 0     // f_backward_pipe: layer-0 gradient sums per lane in registers (1) or reduced per tile (0: 12 registers fewer)
#endif
#ifndef GPE_PIPE_PREFETCH
#define GPE_PIPE_PREFETCH 1  // ... the next tile's top-layer stored jets requested one product phase ahead
#endif
#ifndef GPE_COOP_PRIO_P
#define GPE_COOP_PRIO_P 0    // ... and of its VALU / LDS phases
#endif

// pack hidden-hidden weights (linear maps 1..L-1) in MFMA fragment order.
//   Wpk [j-1][nt][kt][lane][s] = W_j[16nt + (lane&15)][16kt + 4(lane>>4) + s]      (forward A operand)
//   WpkT[j-1][kt][nt][lane][s] = W_j[16nt + 4(lane>>4) + s][16kt + (lane&15)]      (backward A operand)
// x = hi + mid + lo + O(2^-24 |x|), every piece exactly representable in bf16 (truncation: the remainders are exact in fp32)
GPE_DEV float bf16_trunc(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u); }
GPE_DEV void bf16_split3(float x, float& hi, float& mid, float& lo) {
    hi = bf16_trunc(x);
    const float r1 = x - hi;
    mid = bf16_trunc(r1);
    lo = r1 - mid;                                                  // (its low 16 bits are dropped when it is packed)
}
GPE_DEV void pack_weight_element(const NetDesc& nd, int H, const float* __restrict__ theta, float* __restrict__ Wpk,
                                 float* __restrict__ WpkT, int idx) {
    const int NT = H / 16;
    int per = H * H;
    int j = idx / per + 1, e = idx % per;
    int s = e & 3, lane = (e >> 2) & 63, t2 = e >> 8;   // t2 = nt*NT+kt (or kt*NT+nt)
    int a = t2 / NT, b = t2 % NT;
    int i = lane & 15, q = lane >> 4;
    const float* W = theta + nd.offW[j];
    Wpk[idx] = W[(16 * a + i) * H + 16 * b + 4 * q + s];           // a = nt, b = kt
    WpkT[idx] = W[(16 * b + 4 * q + s) * H + 16 * a + i];          // a = kt, b = nt
    // bf16 pieces for f_forward_b6 (behind WpkT in the same allocation):
    //   W6[j-1][piece][nt][kb][lane][e] = piece of W_j[16nt + (lane&15)][32kb + 16(e>>2) + 4(lane>>4) + (e&3)],  W = hi + mid + lo
    if (H <= 64) {                                                  // (the kernels that read the pieces exist for H <= 64)
        unsigned short* W6 = reinterpret_cast<unsigned short*>(WpkT + (size_t)(nd.n_lin - 2) * per);
        const int e8 = e & 7, ln = (e >> 3) & 63, t3 = e >> 9;     // t3 = nt * (NT/2) + kb
        const int nt = t3 / (NT / 2), kb = t3 % (NT / 2);
        const float wv = W[(16 * nt + (ln & 15)) * H + 32 * kb + 16 * (e8 >> 2) + 4 * (ln >> 4) + (e8 & 3)];
        float ph, pm, pl;
        bf16_split3(wv, ph, pm, pl);
        const size_t o = ((size_t)(j - 1) * 3 * per) + (size_t)e;  // piece stride = per
        W6[o] = (unsigned short)(__builtin_bit_cast(unsigned, ph) >> 16);
        W6[o + per] = (unsigned short)(__builtin_bit_cast(unsigned, pm) >> 16);
        W6[o + 2 * (size_t)per] = (unsigned short)(__builtin_bit_cast(unsigned, pl) >> 16);
        // ... and of the transposed maps for the adjoint products of f_backward_coop<..., B6> (behind W6):
        //   WT6[j-1][piece][kt][kb][lane][e] = piece of W_j[32kb + 16(e>>2) + 4(lane>>4) + (e&3)][16kt + (lane&15)]     (t3 = kt * (NT/2) + kb)
        unsigned short* WT6 = W6 + (size_t)(nd.n_lin - 2) * 3 * per;
        const float wt = W[(32 * kb + 16 * (e8 >> 2) + 4 * (ln >> 4) + (e8 & 3)) * H + 16 * nt + (ln & 15)];
        bf16_split3(wt, ph, pm, pl);
        WT6[o] = (unsigned short)(__builtin_bit_cast(unsigned, ph) >> 16);
        WT6[o + per] = (unsigned short)(__builtin_bit_cast(unsigned, pm) >> 16);
        WT6[o + 2 * (size_t)per] = (unsigned short)(__builtin_bit_cast(unsigned, pl) >> 16);
    }
}
__global__ void k_pack_weights(NetDesc nd, int H, const float* __restrict__ theta, float* __restrict__ Wpk,
                               float* __restrict__ WpkT) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < (nd.n_lin - 2) * H * H) pack_weight_element(nd, H, theta, Wpk, WpkT, idx);
}
// First kernel of a step: zero the step's accumulators (double sums, gradient + exchange tail, boundary-batch gradient) and,
// on the fused path after a parameter update, repack the hidden-hidden weights -- one launch instead of three fills + a pack.
__global__ void k_begin(NetDesc nd, int H, const float* __restrict__ theta, float* __restrict__ Wpk, float* __restrict__ WpkT,
                        int n_pack, double* __restrict__ dbl, int n_dbl, float* __restrict__ grad, int n_grad,
                        float* __restrict__ grad_bc, int n_bc) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < n_pack) pack_weight_element(nd, H, theta, Wpk, WpkT, idx);
    if (idx < n_dbl) dbl[idx] = 0.0;
    if (idx < n_grad) grad[idx] = 0.f;
    if (idx < n_bc) grad_bc[idx] = 0.f;
}

// WLDS: the packed hidden-hidden weights ((L-1)*H*H floats) are staged once per workgroup into LDS and the MFMA A
// operands are read from there (ds_read_b128, ~100 cycles) instead of from L2 (~600 cycles) right before each use.
// HEADF (whole steps of real psi without orthogonality / Riesz / symmetry terms, batches beyond the cooperative kernel's range): the wave
// also runs the head of its tile's rows (head_point_real on the q = 0 lanes, which hold the point's output jets after the reduction) and
// the workgroup leaves one (num, den, bse) triple in ha.slots -- k_head_pde is not launched (36.8 us of the NS step), the consumer
// (k_seed_pde, or the reverse kernel that forms the seeds itself) adds the triples in a fixed order.
template <int H, int C, int E, int NOUT, bool WLDS, bool HEADF = false>
__global__ __launch_bounds__(256, ((C <= 5 && H <= 64) ? GPE_FWD_WAVES : 1)) void f_forward(NetDesc nd, const float* __restrict__ theta,
                                                                 const float* __restrict__ Wpk,
                                                                 Pts x, float* __restrict__ stored,
                                                                 float* __restrict__ O, int64_t N, int64_t ld,
                                                                 int store_acts, int old_share_q10, HeadArgs ha) {
    static_assert(!HEADF || NOUT == 1, "head in the forward kernel: real psi");
    double hnum = 0.0, hden = 0.0, hbse = 0.0;                   // HEADF: this lane's partial sums (q = 0 lanes)
    constexpr int D = C - 1 - E, NT = H / 16, NF = NT * 4;
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    const unsigned lane4 = (unsigned)lane * 4u;
    const int L = nd.n_lin - 1;
    const int dim = nd.dim;
    const int64_t ntiles = (N + 15) >> 4;
    // wave index through readfirstlane: the tile number (and with it every stored-activation address) lives in SGPRs, the
    // address arithmetic runs on the scalar unit instead of taking VALU cycles from the matrix products
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float shift = nd.shift;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float* w0s = lds_f;
    float* lds_w = lds_f + ((small_count(nd, H) + 3) & ~3);
    stage_layer0<H>(w0s, theta, nd, 256);
    const buf_t rW = buf_make(Wpk, (unsigned)((L - 1) * H * H * 4));
    if constexpr (WLDS) {
        stage_copy16(reinterpret_cast<f32x4*>(lds_w), reinterpret_cast<const f32x4*>(Wpk), (L - 1) * H * H / 4, (int)threadIdx.x, 256);
    }
    __syncthreads();

    // point coordinates of the first tile; inside the loop the NEXT tile's are requested before this tile's math
    // tiles of this wave: wave0, wave0 + nwaves, ... -- or, with old_share_q10 set (large batches on two workgroups per CU: workgroups
    // b and b + G/2 share a CU, and the waves of the first-dispatched one win every arbitration), the tiles of such a pair of waves split
    // unevenly so that both finish together (the same device as f_backward_pipe's, SeedArgs::old_share_q10)
    int64_t tile0 = wave0, tstep = nwaves, tend = ntiles;
    if (old_share_q10 > 0 && (gridDim.x & 1) == 0) {
        const int64_t half = nwaves >> 1, p = wave0 % half;
        const int64_t cnt = p < ntiles ? (ntiles - p + half - 1) / half : 0;
        const int64_t n_old = (cnt * old_share_q10 + 512) >> 10;
        tstep = half;
        if (wave0 < half) { tile0 = p; tend = p + n_old * half < ntiles ? p + n_old * half : ntiles; }
        else tile0 = p + n_old * half;
    }
    float xn[3] = {0.f, 0.f, 0.f};
    if (tile0 < tend) {
        const int64_t p0 = min(tile0 * 16 + m, N - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k < dim) xn[k] = pts_at(x, p0, dim, k);
    }
    for (int64_t tile = tile0; tile < tend; tile += tstep) {
        const int64_t pm = tile * 16 + m;
        const bool valid = pm < N;
        float xv[3] = {xn[0], xn[1], xn[2]};
        if (tile + tstep < tend) {
            const int64_t pn = min((tile + tstep) * 16 + m, N - 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) if (k < dim) xn[k] = pts_at(x, pn, dim, k);
        }

        // this tile's block of stored activations: [L-1][C][NT][256] floats
        const buf_t rS = buf_make(stored + (size_t)tile * (L - 1) * C * NT * 256, (unsigned)((L - 1) * C * NT * 1024));
        float bufA[C][NF], bufB[C][NF];
        // ---- layer 0 (K = dim <= 3): VALU -----------------------------------------------------------
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 st[C];
            layer0_st<H, C, E>(w0s, xv, nt, q, st);
            f32x4 a4[C];
            act_from_stored<D, E>(st[0], st + 1, st + 1 + D, shift, a4);      // layer0_st leaves the second-order channels zero
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) bufA[c][nt * 4 + r] = a4[c][r];
        }
        // ---- hidden -> hidden layer j on the matrix cores: a_in -> a_out -------------------------------------
        auto layer = [&](const float (&a_in)[C][NF], float (&a_out)[C][NF], int j) {
            const float* Wp = (WLDS ? (const float*)lds_w : Wpk) + (size_t)(j - 1) * H * H;
            const float* bj = w0s + (4 + (j - 1)) * H;          // LDS copy of b_j
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 w[NT];
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    if constexpr (WLDS) w[kt] = *reinterpret_cast<const f32x4*>(Wp + (nt * NT + kt) * 256 + lane4);
                    else w[kt] = buf_load4(rW, lane4 * 4u, (unsigned)(((j - 1) * NT * NT + nt * NT + kt) * 1024));
                }
                f32x4 acc[C];
                acc[0] = *reinterpret_cast<const f32x4*>(&bj[16 * nt + 4 * q]);
#pragma unroll
                for (int c = 1; c < C; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[kt][s], a_in[c][kt * 4 + s], acc[c], 0, 0, 0);

                const f32x4 tt = gpe_tanh(acc[0]);
                f32x4 a4[C];
                act_from_stored<D, E>(tt, acc + 1, acc + 1 + D, shift, a4);
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a_out[c][nt * 4 + r] = a4[c][r];
                if (store_acts) {                                    // uniform base (SGPRs) + 32-bit lane offset
                    const unsigned so = (unsigned)(((j - 1) * C * NT + nt) * 1024);
                    buf_store4(tt, rS, lane4 * 4u, so);
#pragma unroll
                    for (int c = 1; c < C; ++c) buf_store4(acc[c], rS, lane4 * 4u, so + (unsigned)(c * NT * 1024));
                }
            }
        };
        // ---- output layer (n_out <= 2): VALU dot + reduction over the 4 q-lanes of a point ------------------
        auto output = [&](const float (&a_in)[C][NF]) {
            const float* Wo = w0s + (4 + L - 1) * H;            // LDS copies of W_out, b_out
            const float* bo = w0s + (4 + L - 1 + NOUT) * H;
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
                    if constexpr (HEADF) part[c] = v;
                }
                if constexpr (HEADF) {
                    if (q == 0 && valid) head_point_real<C, E>(ha, xv, pm, N, part, hnum, hden, hbse);
                }
            }
        };
        int j = 1;
        for (; j + 1 < L; j += 2) { layer(bufA, bufB, j); layer(bufB, bufA, j + 1); }   // ping-pong: no register copies
        if (j < L) { layer(bufA, bufB, j); output(bufB); }
        else output(bufA);
    }
    if constexpr (HEADF) {                                       // lanes -> wave -> workgroup, each in a fixed order
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            hnum += __shfl_xor(hnum, o, 64);
            hden += __shfl_xor(hden, o, 64);
            hbse += __shfl_xor(hbse, o, 64);
        }
        __syncthreads();                                         // every wave is done with the LDS copies of the small parameters
        double* hred = reinterpret_cast<double*>(lds_f);
        const int wv = threadIdx.x >> 6;
        if (lane == 0) { hred[wv * 3 + 0] = hnum; hred[wv * 3 + 1] = hden; hred[wv * 3 + 2] = hbse; }
        __syncthreads();
        if (threadIdx.x < 3) {
            double t = 0.0;
            for (int k = 0; k < 4; ++k) t += hred[k * 3 + threadIdx.x];
            ha.slots[(size_t)blockIdx.x * 4 + threadIdx.x] = t;
        }
    }
}

// f_forward with the H x H maps on the bf16 matrix instruction at fp32 accuracy: each fp32 operand is three bf16 pieces
// (hi + mid + lo, 24 significant bits) and a product is the six piece products hh, hm, mh, hl, lh, mm accumulated in fp32 -- the dropped
// ones are O(2^-24).  v_mfma_f32_16x16x32_bf16 covers K = 32 in 16 cycles, so six of them cost 96 cycles where eight
// v_mfma_f32_16x16x4_f32 cost 256 (tools/ubench/mfma_bf16_split.hip: 2.6x, max error 1.4e-6 against 2.1e-6 of the fp32 instruction at
// K = 64).  K slot 8 kq + e of block kb is feature 32 kb + 16 (e >> 2) + 4 kq + (e & 3): a lane's B operand is eight values it
// already holds (two output fragments of the previous layer), so the layers still chain without data movement; the weights are
// packed to match (pack_weight_element).  Price: ~5 VALU instructions per activation value for the split.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
GPE_DEV unsigned bf16_pack2(float lo_elem, float hi_elem) {         // bf16(lo_elem) | bf16(hi_elem) << 16, by truncation
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi_elem), __builtin_bit_cast(unsigned, lo_elem), 0x07060302u);
}
template <int H, int C, int E, int NOUT, bool WLDS>
__global__ __launch_bounds__(256, 2) void f_forward_b6(NetDesc nd, const float* __restrict__ theta, const float* __restrict__ WpkT,
                                                        Pts x, float* __restrict__ stored, float* __restrict__ O, int64_t N,
                                                        int64_t ld, int store_acts) {
    constexpr int D = C - 1 - E, NT = H / 16, NF = NT * 4, KB = NT / 2;
    static_assert(NT % 2 == 0, "K blocks of 32 features");
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    const unsigned lane16 = (unsigned)lane * 16u;
    const int L = nd.n_lin - 1;
    const int dim = nd.dim;
    const int64_t ntiles = (N + 15) >> 4;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float shift = nd.shift;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float* w0s = lds_f;
    stage_layer0<H>(w0s, theta, nd, 256);
    // the piece arrays: [map][piece][nt][kb][lane] x 16 B; WLDS: staged once per workgroup (72 KB for three 64 x 64 maps)
    const float* W6 = WpkT + (size_t)(L - 1) * H * H;
    const buf_t rW = buf_make(W6, (unsigned)((L - 1) * 3 * H * H * 2));
    const u32x4* lds_w = reinterpret_cast<const u32x4*>(lds_f + ((small_count(nd, H) + 3) & ~3));
    if constexpr (WLDS) {
        u32x4* dst = reinterpret_cast<u32x4*>(lds_f + ((small_count(nd, H) + 3) & ~3));
        stage_copy16(dst, reinterpret_cast<const u32x4*>(W6), (L - 1) * 3 * H * H * 2 / 16, (int)threadIdx.x, 256);
    }
    __syncthreads();

    float xn[3] = {0.f, 0.f, 0.f};
    if (wave0 < ntiles) {
        const int64_t p0 = min(wave0 * 16 + m, N - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k < dim) xn[k] = pts_at(x, p0, dim, k);
    }
    for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
        const int64_t pm = tile * 16 + m;
        const bool valid = pm < N;
        float xv[3] = {xn[0], xn[1], xn[2]};
        if (tile + nwaves < ntiles) {
            const int64_t pn = min((tile + nwaves) * 16 + m, N - 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) if (k < dim) xn[k] = pts_at(x, pn, dim, k);
        }
        const buf_t rS = buf_make(stored + (size_t)tile * (L - 1) * C * NT * 256, (unsigned)((L - 1) * C * NT * 1024));
        float act[C][NF];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {                           // layer 0 (K = dim <= 3): VALU
            f32x4 st[C];
            layer0_st<H, C, E>(w0s, xv, nt, q, st);
            f32x4 a4[C];
            act_from_stored<D, E>(st[0], st + 1, st + 1 + D, shift, a4);
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) act[c][nt * 4 + r] = a4[c][r];
        }
        for (int j = 1; j < L; ++j) {
            // ---- split the layer input: B operand pieces, eight k slots per lane and K block --------------------------------
            u32x4 bp[C][KB][3];
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr) {                // pair pr = slots 2pr, 2pr+1 = registers (2kb + (pr>>1))*4 + 2(pr&1), +1
                        const float x0 = act[c][(2 * kb + (pr >> 1)) * 4 + 2 * (pr & 1)], x1 = act[c][(2 * kb + (pr >> 1)) * 4 + 2 * (pr & 1) + 1];
                        float h0, m0, l0, h1, m1, l1;
                        bf16_split3(x0, h0, m0, l0);
                        bf16_split3(x1, h1, m1, l1);
                        bp[c][kb][0][pr] = bf16_pack2(x0, x1);      // (the top halves of x and of its hi piece are the same bits)
                        bp[c][kb][1][pr] = bf16_pack2(m0, m1);
                        bp[c][kb][2][pr] = bf16_pack2(l0, l1);
                    }
            const float* bj = w0s + (4 + (j - 1)) * H;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                u32x4 wp[KB][3];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc)
                        wp[kb][pc] = WLDS ? lds_w[((((j - 1) * 3 + pc) * NT + nt) * KB + kb) * 64 + lane]
                                          : __builtin_bit_cast(u32x4, buf_load4(rW, lane16, (unsigned)(((((j - 1) * 3 + pc) * NT + nt) * KB + kb) * 1024)));
                f32x4 acc[C];
                acc[0] = *reinterpret_cast<const f32x4*>(&bj[16 * nt + 4 * q]);
#pragma unroll
                for (int c = 1; c < C; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                // smallest products first; C independent accumulator chains
                constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wp[kb][PA[t]]),
                                                                             __builtin_bit_cast(bf16x8, bp[c][kb][PB[t]]), acc[c], 0, 0, 0);
                const f32x4 tt = gpe_tanh(acc[0]);
                f32x4 a4[C];
                act_from_stored<D, E>(tt, acc + 1, acc + 1 + D, shift, a4);
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) act[c][nt * 4 + r] = a4[c][r];     // (the pieces of the old values are already in bp)
                if (store_acts) {
                    const unsigned so = (unsigned)(((j - 1) * C * NT + nt) * 1024);
                    buf_store4(tt, rS, lane16, so);
#pragma unroll
                    for (int c = 1; c < C; ++c) buf_store4(acc[c], rS, lane16, so + (unsigned)(c * NT * 1024));
                }
            }
        }
        // ---- output layer (n_out <= 2): VALU dot + reduction over the 4 q-lanes of a point ------------------
        const float* Wo = w0s + (4 + L - 1) * H;
        const float* bo = w0s + (4 + L - 1 + NOUT) * H;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float part[C];
#pragma unroll
            for (int c = 0; c < C; ++c) part[c] = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(&Wo[o * H + 16 * nt + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < C; ++c) part[c] = fmaf(wv[r], act[c][nt * 4 + r], part[c]);
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

// ---- diagnostic build only (-DGPE_STAMP): per-phase cycle shares of the reverse kernel via s_memtime ---------------
__device__ unsigned long long g_stamps[16];
#ifdef GPE_STAMP
GPE_DEV unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(i) do { unsigned long long _t = stamp_now(); st_acc[i] += _t - st_last; st_last = _t; } while (0)
// f_backward_pipe: phase sums per wave (g_stamps[4 w + phase]: 0 barrier wait, 1 products, 2 VALU / LDS part) and, for every
// workgroup, the stamps of tile iterations PT_IT0, PT_IT0 + 1 (g_trace[block][wave][0] = HW_ID, [1 + 12 it + 4 k + e]: interval k,
// event e = 0 at the barrier, 1 behind it, 2 behind the products, 3 behind the VALU part)
#define PT_IT0 40
__device__ unsigned long long g_trace[512 * 4 * 32];
#define PSTAMP(ph, ev) do { unsigned long long _t = stamp_now(); if ((ph) >= 0) st_acc[(ph) >= 0 ? (ph) : 0] += _t - st_last; st_last = _t; \
        if (lane == 0 && vb < 512 && (it == PT_IT0 || it == PT_IT0 + 1))                                                       \
            g_trace[(vb * 4 + w) * 32 + 1 + 12 * (it - PT_IT0) + (ev)] = _t; } while (0)
#else
#define STAMP(i) do { } while (0)
#define PSTAMP(ph, ev) do { } while (0)
#endif

// Reverse pass.  Ob = dLoss/dO ([C][NOUT][ld]).  gslab: [gridDim.x][Ppad] per-workgroup gradient slabs.
// Dynamic LDS: gacc[Ppad] | g0[4H] (layer-0 gradients, padded) | nwaves x C transposition tiles | w0s[4H] | (WLDS) W^T.
// WLDS: 512-thread workgroups (one per CU) that also keep the packed transposed weights in LDS.
// NHH > 0 ("register accumulation"): the number of hidden-hidden maps is the compile-time constant NHH, the kernel runs
// one wave per SIMD (256-thread workgroups, up to 512 registers per lane) and every wave keeps its share of ALL the
// H x H weight gradients in MFMA accumulators across all the tiles it processes -- the products dW += Zb X^T chain
// straight into them and LDS float atomics (measured at ~0.4 lane-adds per clock per CU: the limiter of the NHH = 0
// variant) are used only for the small parameters and once per wave at the end.
// H > 64 ("GACC"): the parameter vector no longer fits LDS (266 KB for [2,128x5,1]); gradients are accumulated with
// global float atomics into one of `nslab` L2-resident slabs (blockIdx % nslab) that the host zeroes before the launch --
// 16 KB of atomic traffic per point for cfg3, well inside the chip's ~1.3 TB/s atomic rate at the kernel's compute rate.
template <int H, int C, int E, int NOUT, bool WLDS, int NHH>
__global__ __launch_bounds__((NHH > 0 ? 256 : (WLDS ? 512 : 256)), ((NHH > 0 || H > 64) ? 1 : (C <= 5 ? GPE_BWD_WAVES : 1))) void f_backward(NetDesc nd, const float* __restrict__ theta,
                                                                  const float* __restrict__ WpkT,
                                                                  Pts x,
                                                                  const float* __restrict__ stored,
                                                                  const float* __restrict__ Ob, float* __restrict__ gslab,
                                                                  int64_t N, int64_t ld, int Ppad, int nslab) {
    constexpr int D = C - 1 - E, NT = H / 16, NF = NT * 4;
    constexpr int NTHR = (NHH > 0) ? 256 : (WLDS ? 512 : 256);
    constexpr bool RACC = NHH > 0;
    constexpr bool GACC = H > 64;
    static_assert(!RACC || WLDS, "register accumulation variant keeps W^T in LDS");
    static_assert(!GACC || (!RACC && !WLDS), "wide layers: global accumulation, weights from L2");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* gacc = GACC ? gslab + (size_t)(blockIdx.x % nslab) * Ppad : lds;
    float* g0 = GACC ? lds : lds + Ppad;
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    const int wib = threadIdx.x >> 6;
    float* TT = g0 + 4 * H + wib * (C * F_TILE);      // C transposition tiles, private to this wave
    float* w0s = g0 + 4 * H + (NTHR / 64) * (C * F_TILE);
    float* lds_w = w0s + ((small_count(nd, H) + 3) & ~3);
    const int L = nd.n_lin - 1;
    const int dim = nd.dim;
    const float shift = nd.shift;
    const int64_t ntiles = (N + 15) >> 4;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float* Wo = w0s + (4 + L - 1) * H;                // LDS copy of W_out

    if constexpr (GACC) { for (int i = threadIdx.x; i < 4 * H; i += NTHR) g0[i] = 0.f; }
    else { for (int i = threadIdx.x; i < Ppad + 4 * H; i += NTHR) gacc[i] = 0.f; }      // gacc and g0 are contiguous
    stage_layer0<H>(w0s, theta, nd, NTHR);
    if constexpr (WLDS) {
        stage_copy16(reinterpret_cast<f32x4*>(lds_w), reinterpret_cast<const f32x4*>(WpkT), (L - 1) * H * H / 4, (int)threadIdx.x, NTHR);
    }
    __syncthreads();

    f32x4 dwacc[RACC ? NHH : 1][NT][NT];
    if constexpr (RACC) {
#pragma unroll
        for (int a = 0; a < NHH; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int c = 0; c < NT; ++c) dwacc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#ifdef GPE_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = stamp_now();
#endif
    for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
        const int64_t pm = tile * 16 + m;
        const bool valid = pm < N;
        const int64_t pl = valid ? pm : N - 1;
        float xv[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k < dim) xv[k] = pts_at(x, pl, dim, k);
        float ob[NOUT][C];
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int c = 0; c < C; ++c) ob[o][c] = valid ? Ob[((int64_t)c * NOUT + o) * ld + pm] : 0.f;

        // stored (t, z_k, z_kk) of hidden layer h, feature tile kt: HBM for h >= 1, recomputed from x for h = 0
        auto load_st = [&](int h, int kt, f32x4 (&st)[C]) {
            if (h >= 1) {
                const float* sp = stored + ((((size_t)tile * (L - 1) + (h - 1)) * C) * NT + kt) * 256 + lane * 4;
#pragma unroll
                for (int c = 0; c < C; ++c) st[c] = *reinterpret_cast<const f32x4*>(sp + (size_t)c * NT * 256);
            } else {
                layer0_st<H, C, E>(w0s, xv, kt, q, st);
            }
        };

        // ---- output layer: dWout, dbout, adjoint into the last hidden layer, activation adjoint --------
        float zb[C][NF];
        {
            float gwo[NOUT][NF];                  // per-point contributions to dW_out, reduced over the tile afterwards
            f32x4 stn[C];
            load_st(L - 1, 0, stn);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 st[C];
#pragma unroll
                for (int c = 0; c < C; ++c) st[c] = stn[c];
                if (nt + 1 < NT) load_st(L - 1, nt + 1, stn);            // prefetch: HBM latency behind this tile's math
                f32x4 wo[NOUT];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) wo[o] = *reinterpret_cast<const f32x4*>(&Wo[o * H + 16 * nt + 4 * q]);
                f32x4 a4[C], ab4[C], zv4[C];
                act_from_stored<D, E>(st[0], st + 1, st + 1 + D, shift, a4);
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    f32x4 g = (f32x4)(0.f);
#pragma unroll
                    for (int c = 0; c < C; ++c) g = gpe_fma((f32x4)(ob[o][c]), a4[c], g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) gwo[o][nt * 4 + r] = g[r];
                }
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    f32x4 v = (f32x4)(0.f);
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) v = gpe_fma(wo[o], (f32x4)(ob[o][c]), v);
                    ab4[c] = v;
                }
                act_adjoint<D, E>(st[0], st + 1, st + 1 + D, ab4, zv4);
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) zb[c][nt * 4 + r] = zv4[c][r];
            }
#pragma unroll
            for (int o = 0; o < NOUT; ++o) row_reduce_add<NF>(gwo[o], &gacc[nd.offW[L] + o * H], 1, m, q);
            float gbo[NOUT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) gbo[o] = row_sum16(ob[o][0]);
            if (lane == 0) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) atomicAdd(&gacc[nd.offB[L] + o], gbo[o]);
            }
        }
        STAMP(0);
        // ---- hidden -> hidden linear maps j = L-1 .. 1 ---------------------------------------------------
        const int jtop = RACC ? NHH : (L - 1);
#pragma unroll
        for (int j = jtop; j >= 1; --j) {
            // bias gradient of map j
            row_reduce_add<NF>(zb[0], &gacc[nd.offB[j]], 1, m, q);
            // ---- (1) transpose Zb once: C*NT tiles, kept in registers for the weight-gradient products ------------
            STAMP(1);
            f32x4 zt[NT][C];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 zv[C];
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) zv[c][r] = zb[c][nt * 4 + r];
                tiles_transpose<C>(zv, zt[nt], TT, m, q);
            }
            STAMP(2);
            // ---- (2) B1 in place, one channel at a time:  zb[c] <- W_j^T zb[c]  (adjoint of the input jets, before the
            //      activation adjoint).  Per channel NT independent accumulators; the channel's input registers are dead
            //      once its products are issued, so the result overwrites them: no second 16*C-register array.
            const float* WT = (WLDS ? (const float*)lds_w : WpkT) + (size_t)(j - 1) * H * H;
            // KTG accumulators at a time (2 independent MFMA chains cover the 40-cycle dependent latency at a 32-cycle issue);
            // a channel's outputs go to a small buffer because its input registers stay live until the last group.
            constexpr int KTG = (NT >= 2) ? 2 : 1;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                f32x4 res[NT];
#pragma unroll
                for (int k0 = 0; k0 < NT; k0 += KTG) {
                    f32x4 acc[KTG];
#pragma unroll
                    for (int i = 0; i < KTG; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        f32x4 w[KTG];
#pragma unroll
                        for (int i = 0; i < KTG; ++i)
                            w[i] = *reinterpret_cast<const f32x4*>(&WT[(((k0 + i) * NT + nt) * 64 + lane) * 4]);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int i = 0; i < KTG; ++i)
                                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i][s], zb[c][nt * 4 + s], acc[i], 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < KTG; ++i) res[k0 + i] = acc[i];
                }
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) zb[c][kt * 4 + r] = res[kt][r];
            }
            STAMP(5);
            // ---- (3) one pass over the stored activations of hidden layer j-1 (read ONCE, next tile prefetched):
            //      X recompute -> transpose -> B2 products dW_j += Zb X^T -> LDS accumulate; activation adjoint in place.
            {
                f32x4 stn[C];
                load_st(j - 1, 0, stn);
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    f32x4 st[C], xa[C], xt[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) st[c] = stn[c];
                    if (kt + 1 < NT) load_st(j - 1, kt + 1, stn);
                    {
                        f32x4 ab4[C], zv4[C];
                        act_from_stored<D, E>(st[0], st + 1, st + 1 + D, shift, xa);
#pragma unroll
                        for (int c = 0; c < C; ++c)
#pragma unroll
                            for (int r = 0; r < 4; ++r) ab4[c][r] = zb[c][kt * 4 + r];
                        act_adjoint<D, E>(st[0], st + 1, st + 1 + D, ab4, zv4);
#pragma unroll
                        for (int c = 0; c < C; ++c)
#pragma unroll
                            for (int r = 0; r < 4; ++r) zb[c][kt * 4 + r] = zv4[c][r];
                    }
                    STAMP(7);
                    tiles_transpose<C>(xa, xt, TT, m, q);
                    STAMP(3);
                    if constexpr (RACC) {
                        // chain into the persistent accumulators; NT independent chains interleaved
#pragma unroll
                        for (int c = 0; c < C; ++c)
#pragma unroll
                            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    dwacc[RACC ? j - 1 : 0][nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                        zt[nt][c][s2], xt[c][s2], dwacc[RACC ? j - 1 : 0][nt][kt], 0, 0, 0);
                    } else
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        // two accumulators: v_mfma_f32_16x16x4_f32 has a 40-cycle dependent latency at a 32-cycle issue
                        f32x4 dw0 = (f32x4){0.f, 0.f, 0.f, 0.f}, dw1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            dw0 = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[nt][c][0], xt[c][0], dw0, 0, 0, 0);
                            dw1 = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[nt][c][1], xt[c][1], dw1, 0, 0, 0);
                            dw0 = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[nt][c][2], xt[c][2], dw0, 0, 0, 0);
                            dw1 = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[nt][c][3], xt[c][3], dw1, 0, 0, 0);
                        }
                        // LDS accumulate; columns XOR-swizzled by the row quad so that the 4 rows a wave touches per
                        // instruction fall into different banks (undone when the slab is written)
                        const int col = GACC ? (16 * kt + m) : ((16 * kt + m) ^ (((nt * 4 + q) & (NT - 1)) << 4));
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            atomicAdd(&gacc[nd.offW[j] + (16 * nt + 4 * q + r) * H + col], dw0[r] + dw1[r]);
                    }
                    STAMP(4);
                }
            }
        }
        // ---- linear map 0: z = W0 x + b0, dz/dx_k = W0[:,k]  ->  LDS area g0[4][H] = (dW0[:, 0..2], db0) ------------------------
        row_reduce_add<NF>(zb[0], &g0[3 * H], 1, m, q);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < D || (D == 0 && k < dim)) {
                float v[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    v[f] = zb[0][f] * xv[k];
                    if constexpr (C > 1) { if (k < D) v[f] += zb[(1 + k) < C ? (1 + k) : 0][f]; }
                }
                row_reduce_add<NF>(v, &g0[k * H], 1, m, q);
            }
        }
        STAMP(6);
    }
#ifdef GPE_STAMP
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps[i], st_acc[i]);
#endif
    if constexpr (RACC) {      // once per wave: fold the register accumulators into the workgroup's LDS gradient (swizzled)
#pragma unroll
        for (int a = 0; a < NHH; ++a)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    const int col = (16 * kt + m) ^ (((nt * 4 + q) & (NT - 1)) << 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        atomicAdd(&gacc[nd.offW[a + 1] + (16 * nt + 4 * q + r) * H + col], dwacc[a][nt][kt][r]);
                }
    }
    __syncthreads();
    if constexpr (GACC) {                               // only the padded layer-0 block lives in LDS
        for (int i = threadIdx.x; i < 4 * H; i += NTHR) {
            const int n = i % H, k = i / H;
            if (k == 3) atomicAdd(&gacc[nd.offB[0] + n], g0[i]);
            else if (k < dim) atomicAdd(&gacc[nd.offW[0] + n * dim + k], g0[i]);
        }
        return;
    }
    float* slab = gslab + (size_t)blockIdx.x * Ppad;
    const int first = nd.offW[1];                       // layer-0 parameters occupy [0, (dim+1) H)
    const int hid0 = nd.offW[1], hid1 = nd.offW[L];   // hidden-hidden maps 1..L-1 live in [hid0, hid1): H*H weights + H biases each
    for (int i = first + threadIdx.x; i < Ppad; i += NTHR) {
        int src = i;
        if (i >= hid0 && i < hid1) {
            const int e = (i - hid0) % (H * H + H);
            if (e < H * H) {                                  // weight (n, k): stored at column k ^ swz(n)
                const int n = e / H, k = e % H;
                src = i - k + (k ^ ((((n >> 2)) & (NT - 1)) << 4));
            }
        }
        slab[i] = gacc[src];
    }
    for (int i = threadIdx.x; i < 4 * H; i += NTHR) {
        const int n = i % H, k = i / H;
        if (k == 3) slab[nd.offB[0] + n] = g0[i];
        else if (k < dim) slab[nd.offW[0] + n * dim + k] = g0[i];
    }
}

// ---- cooperative forward kernel ----------------------------------------------------------------------------------------
// Same decomposition as f_backward_coop (below): a workgroup of NT = H/16 waves per 16-point tile, wave w computes the
// features 16w..16w+15 of every hidden layer from its register-resident rows of W_j; the activation-jet fragments are
// all-gathered through a ping-pong LDS buffer (one barrier per layer), the output layer is reduced across the waves in LDS.
// A tile's latency is ~NT times shorter than in f_forward: the kernel of choice when a batch has only a few tiles per wave.
// HEADF (small batches of real psi without orthogonality / Riesz / symmetry terms, single-engine steps): the kernel also runs the head
// of its tiles' rows -- u, H u, boundary seeds (head_point_real, gpe_head.h) -- and leaves ONE (num, den, bse) triple per workgroup in
// ha.slots; k_head_pde is not launched, and the sums become reproducible bit for bit (no atomics: the reverse kernel adds the
// triples in index order).
// RES (round 4): the residual-block network of refine/box_to_gaussian_pinn_simulation.py:52-63,100-130 -- hidden layer 0 = act(lin0 x), then
// NHH / 2 blocks  tanh(lin2(tanh(lin1 a)) + a): the even hidden layers 2, 4, .. add the activation jets of the hidden layer two below (the block
// input) to their pre-activation jets.  A wave's slice of the block input is its own output of two layers ago: kept in registers (C x 4).
// Plain tanh inside the blocks, `shift` on layer 0 only.
template <int H, int C, int E, int NOUT, int NHH, bool HEADF = false, bool RES = false>
__global__ __launch_bounds__(H * 4, 2) void f_forward_coop(NetDesc nd, const float* __restrict__ theta,
                                                           const float* __restrict__ Wpk, Pts x,
                                                           float* __restrict__ stored, float* __restrict__ O, int64_t N,
                                                           int64_t ld, int store_acts, HeadArgs ha) {
    static_assert(!HEADF || NOUT == 1, "head in the forward kernel: real psi");
    static_assert(!RES || ((NHH & 1) == 0 && H <= 64), "residual blocks: two maps each");
    constexpr int D = C - 1 - E, NT = H / 16, NTHR = 64 * NT;
    constexpr int L = NHH + 1;
    extern __shared__ __attribute__((aligned(16))) float lds_c[];
    float* w0s = lds_c;
    float* AB = w0s + ((small_count(nd, H) + 3) & ~3);          // [2][C][NT][256]
    float* OP = AB + 2 * C * NT * 256;                          // [NT][NOUT][C][16]
    float* OF = OP + NT * NOUT * C * 16;                        // HEADF: [C][16] final output jets of the tile
    double hnum = 0.0, hden = 0.0, hbse = 0.0;                  // HEADF: partial sums of this workgroup's rows (threads 0..15)
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4, w = threadIdx.x >> 6;
    const int dim = nd.dim;
    const float shift = nd.shift;
    const int64_t ntiles = (N + 15) >> 4;
    f32x4 wreg[NHH][NT];                                        // rows 16w..16w+15 of W_j, K tile kt
#pragma unroll
    for (int a = 0; a < NHH; ++a)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
            wreg[a][kt] = *reinterpret_cast<const f32x4*>(&Wpk[(size_t)a * H * H + ((w * NT + kt) * 64 + lane) * 4]);
    stage_layer0<H>(w0s, theta, nd, NTHR);                      // (behind the weight requests: its two round trips run beside theirs)
    __syncthreads();
    const float* Wo = w0s + (4 + L - 1) * H;
    const float* bo = w0s + (4 + L - 1 + NOUT) * H;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t pm = tile * 16 + m;
        const bool valid = pm < N;
        const int64_t pl = valid ? pm : N - 1;
        float xv[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k < dim) xv[k] = pts_at(x, pl, dim, k);
        f32x4 a[C];
        {   // layer 0, own slice
            f32x4 st[C];
            layer0_st<H, C, E>(w0s, xv, w, q, st);
            act_from_stored<D, E>(st[0], st + 1, st + 1 + D, shift, a);          // layer0_st leaves the second-order channels zero
        }
        f32x4 asave[RES ? C : 1];                                                   // RES: this wave's slice of the current block's input
        if constexpr (RES) {
#pragma unroll
            for (int c = 0; c < C; ++c) asave[c] = a[c];
        }
#pragma unroll
        for (int j = 1; j <= NHH; ++j) {
            float* buf = AB + (j & 1) * (C * NT * 256);
#pragma unroll
            for (int c = 0; c < C; ++c) *reinterpret_cast<f32x4*>(&buf[(c * NT + w) * 256 + lane * 4]) = a[c];
            __syncthreads();
            f32x4 acc[C];
            acc[0] = *reinterpret_cast<const f32x4*>(&w0s[(4 + (j - 1)) * H + 16 * w + 4 * q]);      // b_j
#pragma unroll
            for (int c = 1; c < C; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (H == 128 && GPE_FCOOP_SWP) {
                // one K tile of lookahead on the B fragments.  The eight waves run this loop in lockstep behind the layer's barrier, and the
                // compiler -- short of registers beside the 128-160 weight registers -- requests a K tile's fragments only when the previous
                // tile's last product has issued and waits for them two products later: both waves of a SIMD then sit in the same LDS round
                // trip, sixteen times per layer (tools/loop_dump.py).  Requested one K tile ahead, a round trip hides behind 16 products.
                f32x4 bfn[C];
#pragma unroll
                for (int c = 0; c < C; ++c) bfn[c] = *reinterpret_cast<const f32x4*>(&buf[(c * NT + 0) * 256 + lane * 4]);
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    f32x4 bf[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) bf[c] = bfn[c];
                    if (kt + 1 < NT) {
#pragma unroll
                        for (int c = 0; c < C; ++c) bfn[c] = *reinterpret_cast<const f32x4*>(&buf[(c * NT + kt + 1) * 256 + lane * 4]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j - 1][kt][s2], bf[c][s2], acc[c], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                if constexpr (H == 128 && GPE_FCOOP_ALT_PRIO) {   // SIMD partners (w, w + 4) take turns at the matrix pipe, one K tile each
                    if (((kt & 1) == 0) == (w < 4)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                }
                f32x4 bf[C];
#pragma unroll
                for (int c = 0; c < C; ++c) bf[c] = *reinterpret_cast<const f32x4*>(&buf[(c * NT + kt) * 256 + lane * 4]);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j - 1][kt][s2], bf[c][s2], acc[c], 0, 0, 0);
            }

            if constexpr (H == 128 && GPE_FCOOP_ALT_PRIO) __builtin_amdgcn_s_setprio(0);
            if constexpr (RES) {
                if ((j & 1) == 0) {                                                 // second map of a block: + the block input
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] += asave[c];
                }
            }
            const f32x4 tt = gpe_tanh(acc[0]);
            act_from_stored<D, E>(tt, acc + 1, acc + 1 + D, RES ? 0.f : shift, a);
            if constexpr (RES) {
                if ((j & 1) == 0) {                                                 // ... whose output is the next block's input
#pragma unroll
                    for (int c = 0; c < C; ++c) asave[c] = a[c];
                }
            }
            if (store_acts) {      // through a per-(tile, layer) buffer descriptor: no 64-bit VALU address arithmetic per store
                const buf_t sb = buf_make(stored + ((size_t)tile * (L - 1) + (j - 1)) * (C * NT * 256), (unsigned)(C * NT * 256 * sizeof(float)));
                const int wu = __builtin_amdgcn_readfirstlane(w);
                buf_store4(tt, sb, (unsigned)lane * 16u, (unsigned)(wu * 1024));
#pragma unroll
                for (int c = 1; c < C; ++c) buf_store4(acc[c], sb, (unsigned)lane * 16u, (unsigned)((c * NT + wu) * 1024));
            }
        }
        // output layer: this slice's part of the dot products, reduced over the 4 q-lanes of a point, then over the waves
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(&Wo[o * H + 16 * w + 4 * q]);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float v = wv[0] * a[c][0];
#pragma unroll
                for (int r = 1; r < 4; ++r) v = fmaf(wv[r], a[c][r], v);
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (q == 0) OP[((w * NOUT + o) * C + c) * 16 + m] = v;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < NOUT * C * 16; i += NTHR) {
            const int pmi = i & 15, oc = i >> 4, c = oc % C, o = oc / C;
            float v = (c == 0) ? bo[o] : 0.f;
#pragma unroll
            for (int ww = 0; ww < NT; ++ww) v += OP[((ww * NOUT + o) * C + c) * 16 + pmi];
            const int64_t p = tile * 16 + pmi;
            if (p < N) O[((int64_t)c * NOUT + o) * ld + p] = v;
            if constexpr (HEADF) OF[c * 16 + pmi] = v;
        }
        if constexpr (HEADF) {
            __syncthreads();
            if (threadIdx.x < 16 && valid) {                        // thread t = point t of the tile (its coordinates are in xv)
                float Oj[C];
#pragma unroll
                for (int c = 0; c < C; ++c) Oj[c] = OF[c * 16 + threadIdx.x];
                head_point_real<C, E>(ha, xv, pm, N, Oj, hnum, hden, hbse);
            }
        }
    }
    if constexpr (HEADF) {
        if (threadIdx.x < 64) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                hnum += __shfl_down(hnum, o, 64);
                hden += __shfl_down(hden, o, 64);
                hbse += __shfl_down(hbse, o, 64);
            }
            if (threadIdx.x == 0) {
                ha.slots[(size_t)blockIdx.x * 4 + 0] = hnum;
                ha.slots[(size_t)blockIdx.x * 4 + 1] = hden;
                ha.slots[(size_t)blockIdx.x * 4 + 2] = hbse;
            }
        }
    }
}

// ---- cooperative reverse kernel ------------------------------------------------------------------------------------------
// One workgroup of NT = H/16 waves works on ONE 16-point tile at a time; wave w owns the features 16w..16w+15 of every hidden
// layer ("feature slice").  Compared with f_backward (one wave = one tile, all features):
//   * a wave keeps only its ROWS of every H x H weight gradient: NHH*NT accumulator tiles (48 registers for NS) instead of
//     NHH*NT*NT (192), and its K-slices of W^T stay in registers for the whole kernel (no LDS weight staging) -- so two
//     workgroups fit a CU (2 waves per SIMD), and the VALU / LDS phases of one hide behind the MFMA phases of the other
//     (a lone wave issues a VALU instruction only every ~8 cycles, tools/ubench/mfma_valu.hip);
//   * a tile's latency drops ~NT-fold, which is what small batches (the reference's 4 000 points = 250 tiles) are bound by.
// Per hidden->hidden map the waves exchange two things through LDS, with one barrier each: the adjoint jets z (point-on-lane
// fragments, B operand of W^T z for every wave) and the transposed activation jets X^T (B operand of dW += Z X^T).
// Elementwise work (recompute, activation adjoint), bias / output / layer-0 gradients are local to a slice.
// Gradient slabs: H x H rows straight from the accumulators (each wave owns its rows: plain stores); small parameters via LDS.

// B6 (opt-in, GPE_BWD_B6=1, H <= 64): the adjoint products W_j^T zbar_j as six bf16 matrix products on three-piece splits (see
// f_forward_b6): every wave publishes the PIECES of its zbar fragment (24 B instead of 16 B per four values), the W^T pieces come
// from L2 (six 1 KiB loads per map, requested before the barrier).  The weight-gradient products contract over the 16 POINTS of a
// tile -- half a K = 32 slab -- and stay on the fp32 instruction.
// RES (round 4): residual-block network (see f_forward_coop).  zbar of an even hidden layer j >= 2 also reaches the adjoint of the block input,
// hidden layer j - 2: the wave keeps its slice of zbar_j (C x 4 registers) and adds it to  W_{j-1}^T zbar_{j-1}  one map later.
template <int H, int C, int E, int NOUT, int NHH, bool B6 = false, bool RES = false>
__global__ __launch_bounds__(H * 4, 2) void f_backward_coop(NetDesc nd, const float* __restrict__ theta,
                                                            const float* __restrict__ WpkT, Pts x,
                                                            const float* __restrict__ stored, const float* __restrict__ Ob,
                                                            float* __restrict__ gslab, int64_t N, int64_t ld, int Ppad) {
    constexpr int D = C - 1 - E, NT = H / 16, NTHR = 64 * NT;
    constexpr int L = NHH + 1;                       // index of the output map; hidden layers 0..L-1
    static_assert(H * 4 == NTHR, "one wave per 16-feature slice");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // LDS: gsm (small-parameter gradients) | g0[4][H] | w0s small operands | ZB[C][NT][256] | XT[C][NT][F_TILE] | TT[NT][C][F_TILE]
    const int n_gsm = (L - 1 + NOUT) * H + 4;        // b_1..b_{L-1} | W_out[NOUT][H] | b_out
    float* gsm = lds;
    float* g0 = gsm + ((n_gsm + 3) & ~3);
    float* w0s = g0 + 4 * H;
    float* ZB = w0s + ((small_count(nd, H) + 3) & ~3);
    constexpr int KB = NT / 2;
    float* XT = ZB + (B6 ? 3 * C * KB * 256 : C * NT * 256);     // B6: [piece][C][KB][lane] x 16 B (two tiles' four bf16 values each)
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* TT = XT + C * NT * F_TILE + w * (C * F_TILE);
    const int dim = nd.dim;
    const float shift = nd.shift;
    const int64_t ntiles = (N + 15) >> 4;
    const float* Wo = w0s + (4 + L - 1) * H;

    for (int i = threadIdx.x; i < ((n_gsm + 3) & ~3) + 4 * H; i += NTHR) gsm[i] = 0.f;      // gsm and g0 are contiguous
    stage_layer0<H>(w0s, theta, nd, NTHR);
    // this wave's K-slices of the transposed weights: A operands of  abar[16w..] = sum_nt W_j^T[16w.., 16nt..] z[16nt..].
    // H <= 64: register-resident for the whole kernel.  H = 128: streamed from L2 in chunks of NTC tiles, the first chunk
    // requested before the barrier that precedes its use.
    static_assert(!B6 || H <= 64, "split-bf16 adjoint products: H <= 64");
    constexpr bool WREG = (H <= 64) && !B6 && NHH <= GPE_COOP_WREG_MAX;
    constexpr int NTC = WREG ? NT : 4;
    const float* wbase = WpkT;
    // B6: W^T pieces [map][piece][kt][kb][lane] x 16 B, behind WpkT and the forward pieces
    const buf_t rW6 = buf_make(reinterpret_cast<const unsigned short*>(WpkT + (size_t)NHH * H * H) + (size_t)NHH * 3 * H * H,
                               (unsigned)(NHH * 3 * H * H * 2));
    auto load_w = [&](int a, int nt) {
        return *reinterpret_cast<const f32x4*>(&wbase[(size_t)a * H * H + ((w * NT + nt) * 64 + lane) * 4]);
    };
    f32x4 wreg[WREG ? NHH : 1][WREG ? NT : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int a = 0; a < NHH; ++a)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wreg[a][nt] = load_w(a, nt);
    }
    f32x4 dwacc[NHH][NT];                              // rows 16w..16w+15 of dW_j, column tile kt
#pragma unroll
    for (int a = 0; a < NHH; ++a)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) dwacc[a][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Small-parameter gradients of this wave's slice (bias rows, output weights, layer-0 rows).  H <= 64 (SREG): summed per LANE over
    // all the tiles the workgroup walks -- one packed add each per tile -- and reduced across the 16 point lanes once, after the
    // loop: reducing per tile cost 10 VALU instructions and an LDS atomic for each of the 5 + NHH sums (NS reverse 1.87 -> 1.82 ms).
    // H = 128 (registers hold NHH x 8 accumulator tiles of the H x H gradients) and the 3D three-map kernel (would spill 54
    // registers) keep the per-tile reduction.
    constexpr bool SREG = (H <= 64) && (C * NHH <= 12);
    f32x4 dbacc[NHH], g0acc[4], gwoacc[NOUT];
    float gboacc[NOUT];
#pragma unroll
    for (int a = 0; a < NHH; ++a) dbacc[a] = (f32x4)(0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) g0acc[k] = (f32x4)(0.f);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) { gwoacc[o] = (f32x4)(0.f); gboacc[o] = 0.f; }
    __syncthreads();

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // streamed weights: keep the (tile-invariant) loads inside the loop -- hoisted, they would occupy NHH*NT*4 registers
        if constexpr (!WREG) asm volatile("" : "+s"(wbase));
        const int64_t pm = tile * 16 + m;
        const bool valid = pm < N;
        const int64_t pl = valid ? pm : N - 1;
        float xv[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k < dim) xv[k] = pts_at(x, pl, dim, k);
        float ob[NOUT][C];
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int c = 0; c < C; ++c) ob[o][c] = valid ? Ob[((int64_t)c * NOUT + o) * ld + pm] : 0.f;
        // stored (t, z_k, z_L) of hidden layer h, this wave's feature slice
        // (buffer addressing: the tile's block [L-1][C][NT][256] behind one descriptor, offsets on the scalar unit)
        const buf_t rS = buf_make(stored + (size_t)tile * (L - 1) * C * NT * 256, (unsigned)((L - 1) * C * NT * 1024));
        auto load_st = [&](int h, f32x4 (&st)[C]) {
            if (h >= 1) {
#pragma unroll
                for (int c = 0; c < C; ++c) st[c] = buf_load4(rS, (unsigned)lane * 16u, (unsigned)((((h - 1) * C + c) * NT + w) * 1024));
            } else {
                layer0_st<H, C, E>(w0s, xv, w, q, st);
            }
        };
        // ---- output layer, own slice ----------------------------------------------------------------------------------
        f32x4 zb[C];
        {
            f32x4 st[C];
            load_st(L - 1, st);
            f32x4 wo[NOUT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) wo[o] = *reinterpret_cast<const f32x4*>(&Wo[o * H + 16 * w + 4 * q]);
            {
                f32x4 a4[C], ab4[C];
                act_from_stored<D, E>(st[0], st + 1, st + 1 + D, RES ? 0.f : shift, a4);        // (RES: plain tanh above layer 0; L - 1 >= 1)
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    f32x4 g = SREG ? gwoacc[o] : (f32x4)(0.f);
#pragma unroll
                    for (int c = 0; c < C; ++c) g = gpe_fma((f32x4)(ob[o][c]), a4[c], g);
                    if constexpr (SREG) {
                        // pin the sum here: left free, the scheduler sinks these adds below the last map of the tile and keeps the
                        // recomputed jets alive (and in scratch) across all the product phases
                        asm volatile("" : "+v"(g));
                        gwoacc[o] = g;
                    } else {
                        const float gv[4] = {g[0], g[1], g[2], g[3]};
                        row_reduce4_add(gv, &gsm[(L - 1 + o) * H + 16 * w], m, q);
                    }
                }
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    f32x4 v = (f32x4)(0.f);
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) v = gpe_fma(wo[o], (f32x4)(ob[o][c]), v);
                    ab4[c] = v;
                }
                act_adjoint<D, E>(st[0], st + 1, st + 1 + D, ab4, zb);
            }
            if constexpr (SREG) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) gboacc[o] += ob[o][0];  // (every q-row of lanes holds the same 16 points)
            } else if (w == 0) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    const float gbo = row_sum16(ob[o][0]);
                    if (lane == 0) atomicAdd(&gsm[(L - 1 + NOUT) * H + o], gbo);
                }
            }
        }
        // ---- hidden -> hidden maps j = NHH .. 1 ---------------------------------------------------------------------------
        f32x4 zsave[RES ? C : 1];                                  // RES: own slice of zbar of the last even layer (a block's second map)
#pragma unroll
        for (int j = NHH; j >= 1; --j) {
            if constexpr (!WREG) __builtin_amdgcn_sched_barrier(0);
            if constexpr (RES) {
                if ((j & 1) == 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c) zsave[c] = zb[c];
                }
            }
            if constexpr (SREG) dbacc[j - 1] += zb[0];             // bias gradient of map j, own slice
            else {
                const float z0[4] = {zb[0][0], zb[0][1], zb[0][2], zb[0][3]};
                row_reduce4_add(z0, &gsm[(j - 1) * H + 16 * w], m, q);
            }
            // own slice of z: transposed copy for the weight-gradient products (registers), fragment copy for everybody (LDS)
            f32x4 zt[C];
            tiles_transpose<C>(zb, zt, TT, m, q);
            u32x4 wp6[B6 ? KB : 1][3];
            if constexpr (B6) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                u32x2* Z6 = reinterpret_cast<u32x2*>(ZB);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float hh[4], mm[4], ll[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) bf16_split3(zb[c][r], hh[r], mm[r], ll[r]);
                    const u32x2 ph = {bf16_pack2(zb[c][0], zb[c][1]), bf16_pack2(zb[c][2], zb[c][3])};
                    const u32x2 pm = {bf16_pack2(mm[0], mm[1]), bf16_pack2(mm[2], mm[3])};
                    const u32x2 pl = {bf16_pack2(ll[0], ll[1]), bf16_pack2(ll[2], ll[3])};
                    // slot [piece][c][kb = w >> 1][lane]: 16 B = this tile pair's eight k slots of the lane; this wave fills half (w & 1)
                    Z6[(((0 * C + c) * KB + (w >> 1)) * 64 + lane) * 2 + (w & 1)] = ph;
                    Z6[(((1 * C + c) * KB + (w >> 1)) * 64 + lane) * 2 + (w & 1)] = pm;
                    Z6[(((2 * C + c) * KB + (w >> 1)) * 64 + lane) * 2 + (w & 1)] = pl;
                }
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc)
                        wp6[kb][pc] = __builtin_bit_cast(u32x4, buf_load4(rW6, (unsigned)lane * 16u,
                                                                          (unsigned)((((((j - 1) * 3 + pc) * NT + w) * KB) + kb) * 1024)));
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) *reinterpret_cast<f32x4*>(&ZB[(c * NT + w) * 256 + lane * 4]) = zb[c];
            }
            f32x4 st[C];
            load_st(j - 1, st);                                   // in flight across the barrier and the products below
            f32x4 wnext[NTC];
            if constexpr (!WREG && !B6) {
#pragma unroll
                for (int i = 0; i < NTC; ++i) wnext[i] = load_w(j - 1, i);
            }
            __syncthreads();
            // abar (own slice) = sum_nt W_j^T[slice, nt] z[nt] : C independent accumulator chains
            // H <= 64: the two waves of a SIMD belong to DIFFERENT workgroups in different phases; a wave in a product phase outranks
            // its partner's VALU / LDS phase, whose instructions would otherwise be interleaved one by one into the MFMA stream
            // (fp32 MFMA and VALU do not co-execute: SQ_VALU_MFMA_COEXEC_CYCLES = 0) -- measured 2.08 -> 1.95 ms on the NS workload
            if constexpr (H <= 64) __builtin_amdgcn_s_setprio(GPE_COOP_PRIO);
            f32x4 acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (B6) {
                const u32x4* Z6 = reinterpret_cast<const u32x4*>(ZB);
                constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};     // smallest products first
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    u32x4 bp[C][3];
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) bp[c][pc] = Z6[((pc * C + c) * KB + kb) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wp6[kb][PA[t]]),
                                                                             __builtin_bit_cast(bf16x8, bp[c][PB[t]]), acc[c], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nt0 = 0; nt0 < (B6 ? 0 : NT); nt0 += NTC) {
                f32x4 wv[NTC];
#pragma unroll
                for (int i = 0; i < NTC; ++i) wv[i] = WREG ? wreg[WREG ? j - 1 : 0][WREG ? nt0 + i : 0] : wnext[i];
                if constexpr (!WREG) {
                    if (nt0 + NTC < NT) {
#pragma unroll
                        for (int i = 0; i < NTC; ++i) wnext[i] = load_w(j - 1, nt0 + NTC + i);
                    }
                }
#pragma unroll
                for (int i = 0; i < NTC; ++i) {
                    f32x4 bf[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) bf[c] = *reinterpret_cast<const f32x4*>(&ZB[(c * NT + nt0 + i) * 256 + lane * 4]);
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][s2], bf[c][s2], acc[c], 0, 0, 0);
                }
            }
            if constexpr (H <= 64) __builtin_amdgcn_s_setprio(GPE_COOP_PRIO_P);
            if constexpr (!WREG) __builtin_amdgcn_sched_barrier(0);      // keep the phases' live ranges apart (256-register budget)
            // recompute X of layer j-1 (own slice), activation adjoint -> z of layer j-1, X^T into the shared buffer
            if constexpr (RES) {
                if ((j & 1) == 1 && j + 1 <= NHH) {                // abar of a block input: + zbar of the block's second map (the skip)
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] += zsave[c];
                }
            }
            f32x4 xa[C];
            act_from_stored<D, E>(st[0], st + 1, st + 1 + D, (RES && j - 1 >= 1) ? 0.f : shift, xa);
            act_adjoint<D, E>(st[0], st + 1, st + 1 + D, acc, zb);
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) XT[(c * NT + w) * F_TILE + (4 * q + r) * F_PITCH + tr_wcol(m, q)] = xa[c][r];
            __syncthreads();
            // dW_j[rows of this slice][all columns] += Z^T X : NT independent accumulator chains
            if constexpr (!WREG) __builtin_amdgcn_sched_barrier(0);
            if constexpr (H <= 64) __builtin_amdgcn_s_setprio(GPE_COOP_PRIO);
            constexpr int KTC = (NT > 4) ? 4 : NT;             // column tiles per chunk: KTC independent accumulator chains
#pragma unroll
            for (int kt0 = 0; kt0 < NT; kt0 += KTC)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    f32x4 xf[KTC];
#pragma unroll
                    for (int i = 0; i < KTC; ++i)
                        xf[i] = *reinterpret_cast<const f32x4*>(&XT[(c * NT + kt0 + i) * F_TILE + tr_roff(m, q)]);
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                        for (int i = 0; i < KTC; ++i)
                            dwacc[j - 1][kt0 + i] =
                                __builtin_amdgcn_mfma_f32_16x16x4f32(zt[c][s2], xf[i][s2], dwacc[j - 1][kt0 + i], 0, 0, 0);
                }
            if constexpr (H <= 64) __builtin_amdgcn_s_setprio(GPE_COOP_PRIO_P);
        }
        // ---- linear map 0, own slice: g0[k][n] (k < dim: dW0[n][k]; k = 3: db0[n]) ---------------------------------------------
        {
            if constexpr (SREG) g0acc[3] += zb[0];
            else {
                const float z0[4] = {zb[0][0], zb[0][1], zb[0][2], zb[0][3]};
                row_reduce4_add(z0, &g0[3 * H + 16 * w], m, q);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < D || (D == 0 && k < dim)) {
                    f32x4 v = zb[0] * xv[k];
                    if constexpr (C > 1) { if (k < D) v += zb[(1 + k) < C ? (1 + k) : 0]; }
                    if constexpr (SREG) g0acc[k] += v;
                    else {
                        const float vv[4] = {v[0], v[1], v[2], v[3]};
                        row_reduce4_add(vv, &g0[k * H + 16 * w], m, q);
                    }
                }
            }
        }
    }
    // ---- the per-lane sums: across the 16 point lanes, into the workgroup's LDS block (zeroed above; every wave owns its rows) --------
    if constexpr (SREG) {
        auto reduce4 = [&](const f32x4& a, float* dst16) {
            const float v[4] = {a[0], a[1], a[2], a[3]};
            row_reduce4_add(v, dst16, m, q);
        };
#pragma unroll
        for (int a = 0; a < NHH; ++a) reduce4(dbacc[a], &gsm[a * H + 16 * w]);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) reduce4(gwoacc[o], &gsm[(L - 1 + o) * H + 16 * w]);
#pragma unroll
        for (int k = 0; k < 4; ++k) reduce4(g0acc[k], &g0[k * H + 16 * w]);
        if (w == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float gbo = row_sum16(gboacc[o]);
                if (lane == 0) atomicAdd(&gsm[(L - 1 + NOUT) * H + o], gbo);
            }
        }
    }
    // ---- slab: H x H rows from the accumulators, the rest from LDS ---------------------------------------------------------
    float* slab = gslab + (size_t)blockIdx.x * Ppad;
#pragma unroll
    for (int a = 0; a < NHH; ++a)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[nd.offW[a + 1] + (16 * w + 4 * q + r) * H + 16 * kt + m] = dwacc[a][kt][r];
    __syncthreads();
    for (int i = threadIdx.x; i < (L - 1) * H; i += NTHR) slab[nd.offB[1 + i / H] + i % H] = gsm[i];
    for (int i = threadIdx.x; i < NOUT * H; i += NTHR) slab[nd.offW[L] + i] = gsm[(L - 1) * H + i];
    for (int i = threadIdx.x; i < NOUT; i += NTHR) slab[nd.offB[L] + i] = gsm[(L - 1 + NOUT) * H + i];
    for (int i = threadIdx.x; i < 4 * H; i += NTHR) {
        const int n = i % H, k = i / H;
        if (k == 3) slab[nd.offB[0] + n] = g0[i];
        else if (k < dim) slab[nd.offW[0] + n * dim + k] = g0[i];
    }
}

// ---- pipelined cooperative reverse kernel (H <= 64; the default reverse kernel there) ---------------------------------------
// Same decomposition as f_backward_coop -- a workgroup of NT = H/16 waves per 16-point tile, wave w owns the feature slice 16w..16w+15,
// its K-slices of every W_j^T and its rows of every dW_j stay in registers -- with ONE workgroup barrier per hidden->hidden map
// instead of two.  The weight-gradient product of map j (dW_j += Z_j X_{j-1}^T) feeds nothing downstream, so it is deferred by one
// barrier interval and runs back to back with the adjoint product of map j-1 (abar_{j-2} = W_{j-1}^T z_{j-1}); across the tile
// boundary the dW_1 product of tile t runs with the first adjoint product of tile t+1.  Every interval is then
//     barrier | 2 x 16 C NT matrix instructions (4096 cycles for NS) | recompute + activation adjoint + publish (VALU / LDS) |
// i.e. NHH barriers per tile instead of 2 NHH, each product phase twice as long: the fixed cost of an interval (LDS latency of the
// first operands, barrier skew between the SIMDs) is paid half as often.  The z and X^T exchange buffers are double-buffered (the
// waves of a workgroup are up to one interval apart).  No transposition scratch and no transposed copy of z in registers: the
// feature-on-lane operand of the deferred product is read back from the wave's OWN slice of the z buffer written two intervals
// earlier (nobody else writes it, and the wave itself only overwrites it later in the same interval) with sixteen ds_read_b32 --
// conflict-free because the 16-byte fragment slots of that buffer are XOR-swizzled (slot of lane (m, q) = (m ^ q) + 16 q: the
// ds_read_b128 lane groups still cover each bank row once, and the 32 lanes of a b32 read group land on 32 distinct banks).
// LDS: 2 x (C NT 256) x 2 x 4 B = 64 KB for NS, two workgroups per CU.
// SEEDF (small batches, real psi without orthogonality / Riesz terms): the seeds dLoss/d(output jets) of the collocation rows are formed
// here from u, H u and the step's global sums (seed_point, gpe_common.h) instead of being read from Ob -- k_seed_pde is not launched.
template <int H, int C, int E, int NOUT, int NHH, bool SEEDF = false>
__global__ __launch_bounds__(H * 4, 2) void f_backward_pipe(NetDesc nd, const float* __restrict__ theta,
                                                            const float* __restrict__ WpkT, Pts x,
                                                            const float* __restrict__ stored, const float* __restrict__ Ob,
                                                            float* __restrict__ gslab, int64_t N, int64_t ld, int Ppad, SeedArgs sa) {
    static_assert(!SEEDF || NOUT == 1, "seeds in the reverse kernel: real psi");
    constexpr int D = C - 1 - E, NT = H / 16, NTHR = 64 * NT;
    constexpr int L = NHH + 1;                       // index of the output map; hidden layers 0..L-1
    constexpr int ZSZ = C * NT * 256;                // floats per exchange buffer (z fragments and X^T tiles alike: F_TILE = 256)
    static_assert(H <= 64 && F_TILE == 256, "register-resident weights; 16 x 16 exchange tiles");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tix = threadIdx.x;
    const int64_t vb = blockIdx.x, nvb = gridDim.x;
    // LDS: gsm (small-parameter gradients) | g0[4][H] | w0s small operands | ZB[2][C][NT][256] | XT[2][C][NT][256]
    const int n_gsm = (L - 1 + NOUT) * H + 4;        // b_1..b_{L-1} | W_out[NOUT][H] | b_out
    float* gsm = lds;
    float* g0 = gsm + ((n_gsm + 3) & ~3);
    float* w0s = g0 + 4 * H;
    float* ZB = w0s + ((small_count(nd, H) + 3) & ~3);
    float* XT = ZB + 2 * ZSZ;
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4, w = __builtin_amdgcn_readfirstlane(tix >> 6);
    const int dim = nd.dim;
    const float shift = nd.shift;
    const int64_t ntiles = (N + 15) >> 4;
    const float* Wo = w0s + (4 + L - 1) * H;
    const int zfrag = 4 * ((m ^ q) + 16 * q);          // float offset of the lane's fragment slot in a 256-float z tile (swizzled)
    // ... and of element (feature m, point 4q+s) of such a tile, s = 0..3: the transposed read of the deferred product
    int ztr[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) ztr[s2] = 4 * (((4 * q + s2) ^ (m >> 2)) + 16 * (m >> 2)) + (m & 3);

    float s_lam = 0.f, s_I = 0.f;
    for (int i = tix; i < ((n_gsm + 3) & ~3) + 4 * H; i += NTHR) gsm[i] = 0.f;      // gsm and g0 are contiguous
    // exchange buffers start as zeros: the first interval's deferred product (no tile before it) then adds nothing -- no special case
    for (int i = tix; i < 4 * ZSZ / 4; i += NTHR) reinterpret_cast<f32x4*>(ZB)[i] = (f32x4)(0.f);
    stage_layer0<H>(w0s, theta, nd, NTHR, tix);
    float xv[3] = {0.f, 0.f, 0.f};
    float ob[NOUT][C];
    double r2acc = 0.0;                                // SEEDF: sum of r^2 over this workgroup's collocation rows (wave 0, q = 0 lanes)
    // coordinates and output-jet adjoints of the lane's point, in two steps: the loads (fetch_point), and -- SEEDF -- the seed arithmetic on
    // what they returned (seed_fetched), which needs lambda and the norm integral
    float pf_u = 0.f, pf_Hu = 0.f, pf_V = 0.f;
    int pf_kind = 0;                                   // SEEDF: 1 = collocation row (seeds formed here), 0 = seeds read / padding
    auto fetch_point = [&](int64_t tile) {
        const int64_t pm = tile * 16 + m;
        const bool valid = pm < N;
        const int64_t pl = valid ? pm : N - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k < dim) xv[k] = pts_at(x, pl, dim, k);
        if constexpr (SEEDF) {
            if (valid && pm < sa.n_pde) {                  // collocation row: seeds from u, H u, lambda, the norm integral
                pf_kind = 1;
                pf_u = sa.u[pm]; pf_Hu = sa.Hu[pm];
                if (sa.ph.potential == GPE_POT_PRECOMPUTED) pf_V = sa.Vpre[pm];
            } else {                                       // boundary row riding in the batch (seeded by the head kernel) / padding
                pf_kind = 0;
#pragma unroll
                for (int c = 0; c < C; ++c) ob[0][c] = valid ? Ob[(int64_t)c * ld + pm] : 0.f;
            }
        } else {
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int c = 0; c < C; ++c) ob[o][c] = valid ? Ob[((int64_t)c * NOUT + o) * ld + pm] : 0.f;
        }
    };
    auto seed_fetched = [&]() {
        if constexpr (SEEDF) {
            if (pf_kind == 1) {
                const float V = sa.ph.potential == GPE_POT_PRECOMPUTED ? pf_V : potential_at(sa.ph, xv, nullptr, 0);
                const float r2 = seed_point<C, E>(sa.ph, xv, V, pf_u, pf_Hu, s_lam, s_I, ob[0]);
                if (w == 0 && q == 0) r2acc += (double)r2;
            }
        }
    };
    auto load_point = [&](int64_t tile) { fetch_point(tile); seed_fetched(); };
    // stored (t, z_k, z_L) of hidden layer h >= 1 of a tile, this wave's slice: the tile's block [L-1][C][NT][256] behind one descriptor
    auto load_st = [&](int64_t tile, int h, f32x4 (&st)[C]) {
        const buf_t rS = buf_make(stored + (size_t)tile * (L - 1) * C * NT * 256, (unsigned)((L - 1) * C * NT * 1024));
#pragma unroll
        for (int c = 0; c < C; ++c) st[c] = buf_load4(rS, (unsigned)lane * 16u, (unsigned)((((h - 1) * C + c) * NT + w) * 1024));
    };
    f32x4 wreg[NHH][NT];                               // K-slices of W_j^T: A operands of abar[16w..] = sum_nt W_j^T[16w.., 16nt..] z[16nt..]
#pragma unroll
    for (int a = 0; a < NHH; ++a)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            wreg[a][nt] = *reinterpret_cast<const f32x4*>(&WpkT[(size_t)a * H * H + ((w * NT + nt) * 64 + lane) * 4]);
    f32x4 dwacc[NHH][NT];                              // rows 16w..16w+15 of dW_j, column tile kt
#pragma unroll
    for (int a = 0; a < NHH; ++a)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) dwacc[a][kt] = (f32x4)(0.f);
    // small-parameter gradients of this wave's slice: per-lane sums over all tiles (see f_backward_coop), or per-tile reductions
    constexpr bool SREG = (C * NHH <= 12);
    constexpr bool PREF = GPE_PIPE_PREFETCH && NOUT == 1;   // (complex psi: twice the output-jet adjoints in flight would spill)
    constexpr bool G0REG = SREG && GPE_PIPE_G0REG;       // layer-0 sums per lane too (12 more registers in 2D)
    f32x4 dbacc[NHH], g0acc[4], gwoacc[NOUT];
    float gboacc[NOUT];
#pragma unroll
    for (int a = 0; a < NHH; ++a) dbacc[a] = (f32x4)(0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) g0acc[k] = (f32x4)(0.f);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) { gwoacc[o] = (f32x4)(0.f); gboacc[o] = 0.f; }
    // the first tile's point data and top-layer stored jets are requested here, beside the weight fragments (one L2 / HBM round trip
    // instead of two before the first product -- a workgroup of a 4 000-point batch has ONE tile)
    // tiles of this workgroup: vb, vb + G, ... -- or, with sa.old_share_q10 set (large batches, grid = two workgroups per CU: b and b + G/2
    // share a CU and the first-dispatched one wins every arbitration, 27 k against 37 k cycles per tile), the tiles p, p + G/2, ... of the
    // PAIR split so that both finish together: the older workgroup takes the first old_share_q10 / 1024 of them
    int64_t tile = vb, tstep = nvb, tend = ntiles;
    if (sa.old_share_q10 > 0 && (nvb & 1) == 0) {
        const int64_t half = nvb >> 1, p = vb % half;
        const int64_t cnt = p < ntiles ? (ntiles - p + half - 1) / half : 0;
        const int64_t n_old = (cnt * sa.old_share_q10 + 512) >> 10;
        tstep = half;
        if (vb < half) { tile = p; tend = p + n_old * half < ntiles ? p + n_old * half : ntiles; }
        else tile = p + n_old * half;
    }
    f32x4 stl0[C];
    if (tile < tend) {
        fetch_point(tile);
        if constexpr (L - 1 >= 1) load_st(tile, L - 1, stl0);
    }
    // lambda and the norm integral: from the step sums, or -- the forward kernel ran the head -- from its per-workgroup triples, which are
    // requested HERE, behind the weight fragments and the first tile's data: one round trip for all of them
    if constexpr (SEEDF) {
        if (sa.slots) {                                   // the forward kernel left per-workgroup (num, den, bse): add them in a FIXED tree
            double* sd = reinterpret_cast<double*>(XT);   // (scratch: three doubles of an exchange buffer, put back to zero below)
            if (tix < 64) {                               // lane l: slots l, l + 64, ... in order; then a butterfly over the 64 lanes
                double sv[8][3];                            // (<= 512 slots: all 24 loads in flight at once, then added in order)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int b = tix + 64 * k;
#pragma unroll
                    for (int i = 0; i < 3; ++i) sv[k][i] = b < sa.nslots ? sa.slots[(size_t)b * 4 + i] : 0.0;
                }
                double t3[3] = {0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int i = 0; i < 3; ++i) t3[i] += sv[k][i];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) t3[i] += __shfl_xor(t3[i], o, 64);
                }
                if (tix == 0) { sd[0] = t3[0]; sd[1] = t3[1]; sd[2] = t3[2]; }
            }
            __syncthreads();
            const double tn = sd[0], td = sd[1], tb = sd[2];
            s_lam = (float)(tn / td);
            s_I = (float)td * sa.ph.dx;
            if (vb == 0 && tix == 0) {                    // ... and file the totals where k_update reads them (a separate boundary batch adds its own)
                sa.sums_out[S_NUM] = tn; sa.sums_out[S_DEN] = td;
                if (tb != 0.0) atomicAdd(&sa.lsums_out[LS_BC_SE2], tb);
            }
            __syncthreads();
            if (tix == 0) { sd[0] = 0.0; sd[1] = 0.0; sd[2] = 0.0; }      // (ordered before the tile loop by the barrier below)
        } else {
            s_lam = (float)(sa.sums[S_NUM] / sa.sums[S_DEN]);
            s_I = (float)sa.sums[S_DEN] * sa.ph.dx;
        }
    }
    if (tile < tend) seed_fetched();
    // the weight fragments have landed before the tile loop is entered, and the compiler knows it: else its wait-count bookkeeping
    // (loop-entry state merged with the back edge) puts vmcnt waits in front of the first product phase of every tile, where
    // they stall the matrix instructions on the stored-activation loads that were only just requested
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
    __syncthreads();

    // ---- pieces of a tile's work ---------------------------------------------------------------------------------------------
    // z of a map's output, own slice, into the write-side buffer: B operand of everybody's adjoint product after the next barrier
    auto publish_z = [&](const f32x4 (&zb)[C], float* zw) {
#pragma unroll
        for (int c = 0; c < C; ++c) *reinterpret_cast<f32x4*>(&zw[(c * NT + w) * 256 + zfrag]) = zb[c];
    };
    // output map of a tile (own slice): its weight / bias gradients and z of the top hidden layer
    auto output_stage = [&](const f32x4 (&st)[C], f32x4 (&zb)[C]) {
        f32x4 wo[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) wo[o] = *reinterpret_cast<const f32x4*>(&Wo[o * H + 16 * w + 4 * q]);
        f32x4 a4[C], ab4[C];
        act_from_stored<D, E>(st[0], st + 1, st + 1 + D, shift, a4);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            f32x4 g = SREG ? gwoacc[o] : (f32x4)(0.f);
#pragma unroll
            for (int c = 0; c < C; ++c) g = gpe_fma((f32x4)(ob[o][c]), a4[c], g);
            if constexpr (SREG) {
                asm volatile("" : "+v"(g));            // pin the sum here (else the recomputed jets stay alive across the product phases)
                gwoacc[o] = g;
            } else {
                const float gv[4] = {g[0], g[1], g[2], g[3]};
                row_reduce4_add(gv, &gsm[(L - 1 + o) * H + 16 * w], m, q);
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            f32x4 v = (f32x4)(0.f);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) v = gpe_fma(wo[o], (f32x4)(ob[o][c]), v);
            ab4[c] = v;
        }
        act_adjoint<D, E>(st[0], st + 1, st + 1 + D, ab4, zb);
        if constexpr (SREG) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) gboacc[o] += ob[o][0];      // (every q-row of lanes holds the same 16 points)
        } else if (w == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float gbo = row_sum16(ob[o][0]);
                if (lane == 0) atomicAdd(&gsm[(L - 1 + NOUT) * H + o], gbo);
            }
        }
    };
    auto bias_sum = [&](const f32x4& z0, int a) {      // bias gradient of map a+1, own slice
        if constexpr (SREG) dbacc[a] += z0;
        else {
            const float v[4] = {z0[0], z0[1], z0[2], z0[3]};
            row_reduce4_add(v, &gsm[a * H + 16 * w], m, q);
        }
    };
    // ---- first tile: output map, z of the top hidden layer into buffer 0 ------------------------------------------------------------
    int par = 0;                                       // exchange buffers the next product phase READS
    f32x4 st[C];                                       // stored jets in flight for the next activation adjoint
    if (tile < tend) {
        f32x4 zb[C];
        if constexpr (L - 1 < 1) layer0_st<H, C, E>(w0s, xv, w, q, stl0);
        output_stage(stl0, zb);
        bias_sum(zb[0], NHH - 1);
        publish_z(zb, ZB);
        if constexpr (NHH - 1 >= 1) load_st(tile, NHH - 1, st);
    }
#ifdef GPE_STAMP
    unsigned long long st_acc[3] = {0, 0, 0};
    unsigned long long st_last = stamp_now();
    int it = -1;
    if (lane == 0 && vb < 512) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_trace[(vb * 4 + w) * 32] = ((unsigned long long)xcc << 32) | hwid;
        g_trace[(vb * 4 + w) * 32 + 25] = st_last;       // tile loop entered
    }
#endif
    bool first_tile = true;
    for (; tile < tend; tile += tstep) {
        const bool have_next = tile + tstep < tend;
        float xv_t[3] = {xv[0], xv[1], xv[2]};         // this tile's coordinates (layer 0 recompute, layer-0 gradients)
#ifdef GPE_STAMP
        ++it;
#endif
#pragma unroll
        for (int j = NHH; j >= 1; --j) {
            const float* zr = ZB + par * ZSZ;          // read side (filled in the previous interval)
            const float* xr = XT + par * ZSZ;
            float* zw = ZB + (par ^ 1) * ZSZ;          // write side; its z tiles are two intervals old until this interval's publish
            float* xw = XT + (par ^ 1) * ZSZ;
            f32x4 stn[C];                              // next tile's top-layer stored jets (requested in the tile's last interval)
            PSTAMP(2, 4 * (NHH - j) + 0);
            __syncthreads();
            PSTAMP(0, 4 * (NHH - j) + 1);
            if (j == 1 && have_next && PREF) {
                load_point(tile + tstep);
                if constexpr (L - 1 >= 1) load_st(tile + tstep, L - 1, stn);
            }
            __builtin_amdgcn_s_setprio(GPE_COOP_PRIO);
            // abar (own slice) = sum_nt W_j^T[slice, nt] z[nt] : C independent accumulator chains
            f32x4 acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = (f32x4)(0.f);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 bf[C];
#pragma unroll
                for (int c = 0; c < C; ++c) bf[c] = *reinterpret_cast<const f32x4*>(&zr[(c * NT + nt) * 256 + zfrag]);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j - 1][nt][s2], bf[c][s2], acc[c], 0, 0, 0);
            }
            // deferred weight-gradient product, dW[rows of this slice][all columns] += Z^T X with NT independent accumulator chains: map
            // j+1 of this tile, or map 1 of the previous tile.
            // Z^T: own slice of the write-side z buffer, two intervals old, read feature-on-lane
            if (j < NHH || !first_tile) {                      // (the very first interval of the workgroup has no product behind it)
                f32x4 (&dw)[NT] = dwacc[j < NHH ? j : 0];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    f32x4 zt;
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) zt[s2] = zw[(c * NT + w) * 256 + ztr[s2]];
                    f32x4 xf[NT];
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) xf[kt] = *reinterpret_cast<const f32x4*>(&xr[(c * NT + kt) * 256 + tr_roff(m, q)]);
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                        for (int kt = 0; kt < NT; ++kt) dw[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[s2], xf[kt][s2], dw[kt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(GPE_COOP_PRIO_P);
            PSTAMP(1, 4 * (NHH - j) + 2);
            // recompute X of layer j-1 (own slice), activation adjoint -> z of layer j-1, X^T into the write-side buffer
            f32x4 sj[C];
            if (j - 1 >= 1) {
#pragma unroll
                for (int c = 0; c < C; ++c) sj[c] = st[c];
            } else layer0_st<H, C, E>(w0s, xv_t, w, q, sj);
            f32x4 xa[C], zb[C];
            act_from_stored<D, E>(sj[0], sj + 1, sj + 1 + D, shift, xa);
            act_adjoint<D, E>(sj[0], sj + 1, sj + 1 + D, acc, zb);
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) xw[(c * NT + w) * 256 + (4 * q + r) * F_PITCH + tr_wcol(m, q)] = xa[c][r];
            if (j > 1) {
                bias_sum(zb[0], j - 2);
                publish_z(zb, zw);
                if (j - 2 >= 1) load_st(tile, j - 2, st);
            } else {
                // ---- linear map 0, own slice: g0[k][n] (k < dim: dW0[n][k]; k = 3: db0[n]) ----------------------------------------
                if constexpr (G0REG) g0acc[3] += zb[0];
                else {
                    const float z0[4] = {zb[0][0], zb[0][1], zb[0][2], zb[0][3]};
                    row_reduce4_add(z0, &g0[3 * H + 16 * w], m, q);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (k < D || (D == 0 && k < dim)) {
                        f32x4 v = zb[0] * xv_t[k];
                        if constexpr (C > 1) { if (k < D) v += zb[(1 + k) < C ? (1 + k) : 0]; }
                        if constexpr (G0REG) g0acc[k] += v;
                        else {
                            const float vv[4] = {v[0], v[1], v[2], v[3]};
                            row_reduce4_add(vv, &g0[k * H + 16 * w], m, q);
                        }
                    }
                }
                // (map 1's X^T is on the write side: its dW product runs behind the next barrier)
                // ---- next tile: output map, z of its top hidden layer --------------------------------------------------------------
                if (have_next) {
                    if (!PREF) {
                        load_point(tile + tstep);
                        if constexpr (L - 1 >= 1) load_st(tile + tstep, L - 1, stn);
                    }
                    f32x4 zn[C];
                    if constexpr (L - 1 >= 1) output_stage(stn, zn);
                    else { f32x4 s0[C]; layer0_st<H, C, E>(w0s, xv, w, q, s0); output_stage(s0, zn); }
                    bias_sum(zn[0], NHH - 1);
                    publish_z(zn, zw);
                    if constexpr (NHH - 1 >= 1) load_st(tile + tstep, NHH - 1, st);
                }
            }
            par ^= 1;
        }
        first_tile = false;
    }
#ifdef GPE_STAMP
    if (lane == 0) for (int i = 0; i < 3; ++i) atomicAdd(&g_stamps[4 * w + i], st_acc[i]);
    if (lane == 0 && vb < 512) g_trace[(vb * 4 + w) * 32 + 26] = stamp_now();      // tile loop left
#endif
    // ---- the last tile's deferred dW_1 product ----------------------------------------------------------------------------------
    __syncthreads();
    // the epilogue's lane arithmetic starts from an opaque copy: hoisted above the tile loop it would sit in registers (or scratch) all along
    int le = lane;
    asm volatile("" : "+v"(le));
    const int me = le & 15, qe = le >> 4;
    {
        const float* zo = ZB + (par ^ 1) * ZSZ;
        const float* xr = XT + par * ZSZ;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            f32x4 zt;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) zt[s2] = zo[(c * NT + w) * 256 + 4 * (((4 * qe + s2) ^ (me >> 2)) + 16 * (me >> 2)) + (me & 3)];
            f32x4 xf[NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) xf[kt] = *reinterpret_cast<const f32x4*>(&xr[(c * NT + kt) * 256 + tr_roff(me, qe)]);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
                    dwacc[0][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[s2], xf[kt][s2], dwacc[0][kt], 0, 0, 0);
        }
    }
    if constexpr (SEEDF) {                              // sum of r^2: lanes 0..15 of wave 0 hold the workgroup's partial sums
        if (w == 0) {
            double t = r2acc;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
            if (le == 0 && t != 0.0) atomicAdd(sa.sum_r2, t);
        }
    }
    // ---- the per-lane sums: across the 16 point lanes, into the workgroup's LDS block (zeroed above; every wave owns its rows) --------
    if constexpr (SREG) {
        auto reduce4 = [&](const f32x4& a, float* dst16) {
            const float v[4] = {a[0], a[1], a[2], a[3]};
            row_reduce4_add(v, dst16, me, qe);
        };
#pragma unroll
        for (int a = 0; a < NHH; ++a) reduce4(dbacc[a], &gsm[a * H + 16 * w]);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) reduce4(gwoacc[o], &gsm[(L - 1 + o) * H + 16 * w]);
        if constexpr (G0REG) {
#pragma unroll
            for (int k = 0; k < 4; ++k) reduce4(g0acc[k], &g0[k * H + 16 * w]);
        }
        if (w == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float gbo = row_sum16(gboacc[o]);
                if (le == 0) atomicAdd(&gsm[(L - 1 + NOUT) * H + o], gbo);
            }
        }
    }
    // ---- slab: H x H rows from the accumulators, the rest from LDS ---------------------------------------------------------
    float* slab = gslab + (size_t)vb * Ppad;
#pragma unroll
    for (int a = 0; a < NHH; ++a)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[nd.offW[a + 1] + (16 * w + 4 * qe + r) * H + 16 * kt + me] = dwacc[a][kt][r];
    __syncthreads();
    for (int i = tix; i < (L - 1) * H; i += NTHR) slab[nd.offB[1 + i / H] + i % H] = gsm[i];
    for (int i = tix; i < NOUT * H; i += NTHR) slab[nd.offW[L] + i] = gsm[(L - 1) * H + i];
    for (int i = tix; i < NOUT; i += NTHR) slab[nd.offB[L] + i] = gsm[(L - 1 + NOUT) * H + i];
    for (int i = tix; i < 4 * H; i += NTHR) {
        const int n = i % H, k = i / H;
        if (k == 3) slab[nd.offB[0] + n] = g0[i];
        else if (k < dim) slab[nd.offW[0] + n * dim + k] = g0[i];
    }
}

// One thread's share of a slab column: gslab[g][i] + gslab[g + 16][i] + ... added IN THAT ORDER.  Eight loads are requested before the
// first add (round 4: the plain loop compiled to load -> s_waitcnt vmcnt(0) -> add -> branch, one L2 / HBM round trip per slab, 32 in a
// row at 512 slabs -- the whole 11 us of the reduction at BASELINE configs[1]); the sequence of additions, and with it every bit of
// the sum, is the plain loop's.
__device__ __forceinline__ float slab_column_sum(const float* __restrict__ gslab, int nslab, int Ppad, int i, int g) {
    const float* p = gslab + (size_t)g * Ppad + i;
    const size_t st = (size_t)16 * Ppad;
    float s = 0.f;
    int b = g;
    for (; b + 16 * 7 < nslab; b += 16 * 8, p += 8 * st) {
        const float v0 = p[0], v1 = p[st], v2 = p[2 * st], v3 = p[3 * st], v4 = p[4 * st], v5 = p[5 * st], v6 = p[6 * st], v7 = p[7 * st];
        s += v0; s += v1; s += v2; s += v3; s += v4; s += v5; s += v6; s += v7;
    }
    for (; b < nslab; b += 16, p += st) s += *p;
    return s;
}

// grad[i] (+)= sum_b gslab[b][i] (+ add[i]).  Block = 64 parameters x 16 slab groups (1024 threads); fixed summation order, so the
// slab sum is deterministic for a given grid.  The last reduction of a step also adds the boundary-batch gradient `add` and
// writes the exchange tail (tail_dsc != NULL): grad[P + GT_SUM_R2], grad[P + GT_MSE_SE2].
__global__ __launch_bounds__(1024) void k_grad_reduce(const float* __restrict__ gslab, int nslab, int Ppad, int P,
                                                       float* __restrict__ grad, const float* __restrict__ add,
                                                       const double* __restrict__ tail_dsc, int assign) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const float s = i < P ? slab_column_sum(gslab, nslab, Ppad, i, g) : 0.f;
    red[g][lane] = s;
    __syncthreads();
    if (g == 0 && i < P) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        if (add) t += add[i];
        grad[i] = assign ? t : grad[i] + t;        // the first reduction of a step assigns: the gradient needs no zeroing pass
    }
    if (tail_dsc && blockIdx.x == 0 && threadIdx.x == 0) {
        grad[P + GT_SUM_R2] = (float)tail_dsc[0];
        grad[P + GT_MSE_SE2] = (float)tail_dsc[2];
    }
}
