// gpe_wide.h -- "wide" jet-MLP kernels for gfx950 (CDNA4): hidden width H = 256 (BASELINE configs[4], [3,256x6,1]) and
// H = 128 with five jet channels (3D), where one H x H weight gradient no longer fits beside the adjoint chain in a
// workgroup's registers.  Same numerics, same stored-activation format and the same packed weights as gpe_fused.h.
//
// Decomposition (every kernel: 512-thread workgroups = 8 waves = 2 per SIMD, one workgroup per CU, persistent over tiles of
// 16 points; the H x H maps run on v_mfma_f32_16x16x4_f32, exact fp32):
//   w_forward    whole network.  Wave w owns the output features [16 w RT, 16 (w+1) RT), RT = H/128 row tiles; the activation
//                jets of a layer are all-gathered through ONE LDS buffer [C][H/16][256] (80 KB at H = 256, C = 5: two
//                barriers per layer); the W rows of the wave stream from L2 (packed fragments, double-buffered chunks of 4
//                K tiles); (t, z_k, z_L) are stored fragment-native exactly as f_forward does.
//   w_bwd_out    output layer as a launch of its own: dW_out, db_out, adjoint + activation adjoint of the last hidden layer ->
//                zbar_{L-1} in HBM ([tile][C][H/16][256], fragment-native).  VALU only.  Kept for GPE_WIDE_TOP=0; by default the
//                topmost w_bwd_map launch does this work itself (template parameter TOP).
//   w_bwd_map    ONE hidden->hidden map j per launch, top down:  zbar_j (HBM) -> abar = W_j^T zbar_j (MFMA), activation
//                adjoint -> zbar_{j-1} (HBM; for j = 1 the layer-0 gradients instead), dW_j += Zbar_j X_{j-1}^T (MFMA) kept in
//                accumulator registers across all tiles of the workgroup.  The 256 x 256 gradient does not fit 512 lanes
//                beside the chain, so the COLUMNS of dW_j are split over NSPLIT = 2 workgroups that share the tiles: each
//                computes abar / zbar_{j-1} / X for its half of the layer-(j-1) features (no duplicated MFMA work, no
//                cross-workgroup synchronisation; only zbar_j is read twice -- placed on the same XCD so the second read is
//                an L2 hit) and owns dW_j[:, its half]: 64 accumulator registers per lane.
// Per tile and map the adjoint jets cross HBM once in each direction (80 KB at H = 256): 240 KB per 10 240 MFMAs, 1.7 TB/s at
// the full matrix rate -- a quarter of what HBM sustains, against the generic set's three launches and 61 KB/point per map.
//
// Replaces K1-K3, K12 of SURVEY 2.3 (refine/harmonic_pinn_simulation.py:121-125,158-172,358) for BASELINE configs[4];
// Laplacian template src/gross_pitaevskii_2D.py:183-195 extended to d = 3.
#pragma once
#include "gpe_mfma_util.h"

// ---- diagnostic build only (-DGPE_STAMP): per-phase cycle shares via s_memtime (wave 0 of every workgroup) -------------------
#ifdef GPE_STAMP
__device__ unsigned long long w_stamps[16];
GPE_DEV unsigned long long w_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ unsigned long long w_trace[2][8][16];     // absolute stamps of one iteration of workgroup 0: [kernel][wave][stamp]
#define WSTAMP(i) do { unsigned long long _t = w_now(); st_acc[(i) & 7] += _t - st_last; st_last = _t;               \
                       if (blockIdx.x == 0 && st_iter == 4 && (threadIdx.x & 63) == 0) w_trace[W_TRACE_K][threadIdx.x >> 6][i] = _t; } while (0)
#define WSTAMP_INIT unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_last = w_now(); int st_iter = 0
#define WSTAMP_ITER (++st_iter)
#define WSTAMP_FLUSH(base) do { if (threadIdx.x == 0) for (int _i = 0; _i < 8; ++_i) atomicAdd(&w_stamps[(base) + _i], st_acc[_i]); } while (0)
#else
#define WSTAMP(i) do { } while (0)
#define WSTAMP_INIT do { } while (0)
#define WSTAMP_FLUSH(base) do { } while (0)
#define WSTAMP_ITER do { } while (0)
#endif

#define W_NW 8            // waves per workgroup
#define W_KC 4            // K tiles per streamed weight chunk (forward)
#define W_KCB 2           // ... in the reverse map kernel (register budget: 256 with the 64-register gradient block)

// ---- forward ---------------------------------------------------------------------------------------------------------------
#define W_TRACE_K 0
template <int H, int C, int E, int NOUT>
__global__ __launch_bounds__(512, 2) void w_forward(NetDesc nd, const float* __restrict__ theta,
                                                    const float* __restrict__ Wpk, Pts x, float* __restrict__ stored,
                                                    float* __restrict__ O, int64_t N, int64_t ld, int store_acts) {
    constexpr int D = C - 1 - E, NT = H / 16, RT = NT / W_NW, NTHR = 64 * W_NW;
    static_assert(NT % W_NW == 0, "H must be a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) float lds_w[];
    float* w0s = lds_w;
    float* AB = w0s + ((small_count(nd, H) + 3) & ~3);          // [C][NT][256]
    float* OP = AB + C * NT * 256;                              // [W_NW][NOUT][C][16]
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4, w = threadIdx.x >> 6;
    const int L = nd.n_lin - 1;
    const int dim = nd.dim;
    const float shift = nd.shift;
    const int64_t ntiles = (N + 15) >> 4;
    stage_layer0<H>(w0s, theta, nd, NTHR);
    __syncthreads();
    const float* Wo = w0s + (4 + L - 1) * H;
    const float* bo = w0s + (4 + L - 1 + NOUT) * H;
    int wofs = 0;                                               // opaque zero: keeps the (tile-invariant) weight loads inside the loop
    WSTAMP_INIT;
    const bool hi_even = w < 4;
    // buffer descriptors for the streamed weight rows and the stored jets (as in w_bwd_map: no per-access VALU address arithmetic)
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const unsigned lane16 = (unsigned)lane * 16u;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("" : "+s"(wofs));                         // (an opaque OFFSET, not an opaque pointer: the loads stay global_load)
        const int64_t pm = tile * 16 + m;
        const bool valid = pm < N;
        const int64_t pl = valid ? pm : N - 1;
        float xv[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k < dim) xv[k] = pts_at(x, pl, dim, k);
        f32x4 a[RT][C];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {                      // layer 0 (K = dim <= 3): VALU, own slice
            f32x4 st[C];
            layer0_st<H, C, E>(w0s, xv, w * RT + rt, q, st);
            act_from_stored<D, E>(st[0], st + 1, st + 1 + D, shift, a[rt]);      // layer0_st leaves the second-order channels zero
        }
        for (int j = 1; j < L; ++j) {
            WSTAMP_ITER;
            const buf_t wbuf = buf_make(Wpk + (size_t)(j - 1) * H * H, (unsigned)(H * H * sizeof(float)));
            auto load_w = [&](int rt, int kt) { return buf_load4(wbuf, lane16, (unsigned)(wofs + ((wu * RT + rt) * NT + kt) * 1024)); };
            f32x4 wn[W_KC][RT];
#pragma unroll
            for (int i = 0; i < W_KC; ++i)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) wn[i][rt] = load_w(rt, i);      // first chunk: in flight across the barriers
            WSTAMP(0);
            __syncthreads();                                    // every wave is done reading AB (previous layer / tile)
            WSTAMP(1);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int c = 0; c < C; ++c) *reinterpret_cast<f32x4*>(&AB[(c * NT + w * RT + rt) * 256 + lane * 4]) = a[rt][c];
            __syncthreads();
            WSTAMP(2);
            f32x4 acc[RT][C];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                acc[rt][0] = *reinterpret_cast<const f32x4*>(&w0s[(4 + (j - 1)) * H + 16 * (w * RT + rt) + 4 * q]);      // b_j
#pragma unroll
                for (int c = 1; c < C; ++c) acc[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k0 = 0; k0 < NT; k0 += W_KC) {
                f32x4 wv[W_KC][RT];
#pragma unroll
                for (int i = 0; i < W_KC; ++i)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) wv[i][rt] = wn[i][rt];
                if (k0 + W_KC < NT) {
#pragma unroll
                    for (int i = 0; i < W_KC; ++i)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) wn[i][rt] = load_w(rt, k0 + W_KC + i);
                    __builtin_amdgcn_sched_barrier(0);          // the next chunk's loads are issued BEFORE this chunk's products
                }
#pragma unroll
                for (int i = 0; i < W_KC; ++i) {
                    // the two waves of a SIMD (w, w + 4) take turns at the matrix pipe, one K tile each: under plain age arbitration
                    // the older wave runs its whole loop first and each wave's operand waits are exposed (measured: -6 %)
                    if ((((k0 + i) & 1) == 0) == hi_even) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                    f32x4 bf[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) bf[c] = *reinterpret_cast<const f32x4*>(&AB[(c * NT + k0 + i) * 256 + lane * 4]);
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int c = 0; c < C; ++c)
                                acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][rt][s2], bf[c][s2], acc[rt][c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            WSTAMP(3);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const f32x4 tt = gpe_tanh(acc[rt][0]);
                act_from_stored<D, E>(tt, acc[rt] + 1, acc[rt] + 1 + D, shift, a[rt]);
                if (store_acts) {
                    const buf_t sb = buf_make(stored + ((size_t)tile * (L - 1) + (j - 1)) * (C * NT * 256), (unsigned)(C * NT * 256 * sizeof(float)));
                    buf_store4(tt, sb, lane16, (unsigned)((wu * RT + rt) * 1024));
#pragma unroll
                    for (int c = 1; c < C; ++c) buf_store4(acc[rt][c], sb, lane16, (unsigned)((c * NT + wu * RT + rt) * 1024));
                }
            }
        }
        // output layer: this wave's part of the dot products, reduced over the 4 q-lanes of a point, then over the waves
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float part[C];
#pragma unroll
            for (int c = 0; c < C; ++c) part[c] = 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(&Wo[o * H + 16 * (w * RT + rt) + 4 * q]);
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[c] = fmaf(wv[r], a[rt][c][r], part[c]);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float v = part[c];
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
            for (int ww = 0; ww < W_NW; ++ww) v += OP[((ww * NOUT + o) * C + c) * 16 + pmi];
            const int64_t p = tile * 16 + pmi;
            if (p < N) O[((int64_t)c * NOUT + o) * ld + p] = v;
        }
        // OP is rewritten only after the two barriers of the next tile's first hidden map (L >= 2)
    }
    WSTAMP_FLUSH(0);
}

// ---- forward, several point tiles per pass (round 4; H = 128) -----------------------------------------------------------------------
// w_forward spends one barrier pair per layer on 128 products per wave at H = 128 (640 at H = 256, where it reaches 0.88 of the peak): here a workgroup takes
// TP tiles (16 TP points) through the network together -- every streamed weight fragment serves TP tiles, a barrier pair covers 128 TP products per wave,
// the activation block of one tile runs between the products of another.  Same numerics, same stored-activation format (per tile) as w_forward.
template <int H, int C, int E, int NOUT, int TP>
__global__ __launch_bounds__(512, 2) void w_forward_mt(NetDesc nd, const float* __restrict__ theta,
                                                       const float* __restrict__ Wpk, Pts x, float* __restrict__ stored,
                                                       float* __restrict__ O, int64_t N, int64_t ld, int store_acts) {
    constexpr int D = C - 1 - E, NT = H / 16, NTHR = 64 * W_NW;
    static_assert(NT == W_NW, "one feature tile per wave (H = 128)");
    extern __shared__ __attribute__((aligned(16))) float lds_w[];
    float* w0s = lds_w;
    float* AB = w0s + ((small_count(nd, H) + 3) & ~3);          // [TP][C][NT][256]
    float* OP = AB + TP * C * NT * 256;                         // [TP][W_NW][NOUT][C][16]
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4, w = threadIdx.x >> 6;
    const int L = nd.n_lin - 1;
    const int dim = nd.dim;
    const float shift = nd.shift;
    const int64_t ntiles = (N + 15) >> 4;
    const int64_t npass = (ntiles + TP - 1) / TP;
    stage_layer0<H>(w0s, theta, nd, NTHR);
    __syncthreads();
    const float* Wo = w0s + (4 + L - 1) * H;
    const float* bo = w0s + (4 + L - 1 + NOUT) * H;
    int wofs = 0;                                               // opaque zero: keeps the (pass-invariant) weight loads inside the loop
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const unsigned lane16 = (unsigned)lane * 16u;

    for (int64_t pass = blockIdx.x; pass < npass; pass += gridDim.x) {
        asm volatile("" : "+s"(wofs));
        const int64_t tile0 = pass * TP;
        f32x4 a[TP][C];
#pragma unroll
        for (int t = 0; t < TP; ++t) {                          // layer 0 (K = dim <= 3): VALU, own feature tile
            const int64_t pm = (tile0 + t) * 16 + m;
            const int64_t pl = pm < N ? pm : N - 1;
            float xv[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; ++k) if (k < dim) xv[k] = pts_at(x, pl, dim, k);
            f32x4 st[C];
            layer0_st<H, C, E>(w0s, xv, w, q, st);
            act_from_stored<D, E>(st[0], st + 1, st + 1 + D, shift, a[t]);
        }
        for (int j = 1; j < L; ++j) {
            const buf_t wbuf = buf_make(Wpk + (size_t)(j - 1) * H * H, (unsigned)(H * H * sizeof(float)));
            auto load_w = [&](int kt) { return buf_load4(wbuf, lane16, (unsigned)(wofs + (wu * NT + kt) * 1024)); };
            f32x4 wn[W_KC];
#pragma unroll
            for (int i = 0; i < W_KC; ++i) wn[i] = load_w(i);   // first chunk: in flight across the barriers
            __syncthreads();                                    // every wave is done reading AB (previous layer / pass)
#pragma unroll
            for (int t = 0; t < TP; ++t)
#pragma unroll
                for (int c = 0; c < C; ++c) *reinterpret_cast<f32x4*>(&AB[((t * C + c) * NT + w) * 256 + lane * 4]) = a[t][c];
            __syncthreads();
            f32x4 acc[TP][C];
            const f32x4 bj = *reinterpret_cast<const f32x4*>(&w0s[(4 + (j - 1)) * H + 16 * w + 4 * q]);
#pragma unroll
            for (int t = 0; t < TP; ++t) {
                acc[t][0] = bj;
#pragma unroll
                for (int c = 1; c < C; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k0 = 0; k0 < NT; k0 += W_KC) {
                f32x4 wv[W_KC];
#pragma unroll
                for (int i = 0; i < W_KC; ++i) wv[i] = wn[i];
                if (k0 + W_KC < NT) {
#pragma unroll
                    for (int i = 0; i < W_KC; ++i) wn[i] = load_w(k0 + W_KC + i);
                    __builtin_amdgcn_sched_barrier(0);          // the next chunk's loads are issued BEFORE this chunk's products
                }
#pragma unroll
                for (int i = 0; i < W_KC; ++i)
#pragma unroll
                    for (int t = 0; t < TP; ++t) {
                        f32x4 bf[C];
#pragma unroll
                        for (int c = 0; c < C; ++c) bf[c] = *reinterpret_cast<const f32x4*>(&AB[((t * C + c) * NT + k0 + i) * 256 + lane * 4]);
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                            for (int c = 0; c < C; ++c)
                                acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][s2], bf[c][s2], acc[t][c], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int t = 0; t < TP; ++t) {
                const f32x4 tt = gpe_tanh(acc[t][0]);
                act_from_stored<D, E>(tt, acc[t] + 1, acc[t] + 1 + D, shift, a[t]);
                if (store_acts && tile0 + t < ntiles) {
                    const buf_t sb = buf_make(stored + ((size_t)(tile0 + t) * (L - 1) + (j - 1)) * (C * NT * 256), (unsigned)(C * NT * 256 * sizeof(float)));
                    buf_store4(tt, sb, lane16, (unsigned)(wu * 1024));
#pragma unroll
                    for (int c = 1; c < C; ++c) buf_store4(acc[t][c], sb, lane16, (unsigned)((c * NT + wu) * 1024));
                }
            }
        }
        // output layer: this wave's part of the dot products, reduced over the 4 q-lanes of a point, then over the waves
#pragma unroll
        for (int t = 0; t < TP; ++t)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(&Wo[o * H + 16 * w + 4 * q]);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float v = wv[0] * a[t][c][0];
#pragma unroll
                    for (int r = 1; r < 4; ++r) v = fmaf(wv[r], a[t][c][r], v);
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (q == 0) OP[(((t * W_NW + w) * NOUT + o) * C + c) * 16 + m] = v;
                }
            }
        __syncthreads();
        for (int i = threadIdx.x; i < TP * NOUT * C * 16; i += NTHR) {
            const int pmi = i & 15, oc = (i >> 4) % (NOUT * C), t = i / (NOUT * C * 16), c = oc % C, o = oc / C;
            float v = (c == 0) ? bo[o] : 0.f;
#pragma unroll
            for (int ww = 0; ww < W_NW; ++ww) v += OP[(((t * W_NW + ww) * NOUT + o) * C + c) * 16 + pmi];
            const int64_t p = (tile0 + t) * 16 + pmi;
            if (p < N) O[((int64_t)c * NOUT + o) * ld + p] = v;
        }
        // OP is rewritten only after the two barriers of the next pass's first hidden map (L >= 2)
    }
}

// workgroup b of a launch of G*NSPLIT workgroups -> (half h, tile group g).  Workgroups are dealt round-robin over the 8 XCDs
// (b and b+8 share one): the NSPLIT workgroups of a tile group are placed on the same XCD so that the second read of zbar_j is
// served by that XCD's L2.  Speed only -- any mapping is correct.
GPE_DEV void w_block_role(int b, int G, int NSPLIT, int& h, int& g) {
    if ((G & 7) == 0) { const int xcd = b & 7, idx = b >> 3; h = idx % NSPLIT; g = (idx / NSPLIT) * 8 + xcd; }
    else { h = b % NSPLIT; g = b / NSPLIT; }
}

// ---- reverse, output layer ---------------------------------------------------------------------------------------------------
// gslab: [G][Ppad]; workgroup (h, g) owns the features [16*8*h, 16*8*(h+1)) of the last hidden layer (wave w: feature tile 8h+w).
template <int H, int C, int E, int NOUT, int NSPLIT>
__global__ __launch_bounds__(512, 2) void w_bwd_out(NetDesc nd, const float* __restrict__ theta, Pts x,
                                                    const float* __restrict__ stored, const float* __restrict__ Ob,
                                                    float* __restrict__ Zout, float* __restrict__ gslab, int64_t N,
                                                    int64_t ld, int Ppad, int G) {
    constexpr int D = C - 1 - E, NT = H / 16, NTHR = 64 * W_NW, KTL = NT / NSPLIT;
    static_assert(KTL == W_NW, "one feature tile per wave");
    extern __shared__ __attribute__((aligned(16))) float lds_o[];
    float* gsm = lds_o;                                           // [NOUT][H] dW_out | [NOUT] db_out (+pad)
    float* w0s = gsm + ((NOUT * H + NOUT + 3) & ~3);
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4, w = threadIdx.x >> 6;
    int h, g;
    w_block_role(blockIdx.x, G, NSPLIT, h, g);
    const int L = nd.n_lin - 1;
    const int dim = nd.dim;
    const float shift = nd.shift;
    const int64_t ntiles = (N + 15) >> 4;
    const int ktile = h * KTL + w;
    for (int i = threadIdx.x; i < ((NOUT * H + NOUT + 3) & ~3); i += NTHR) gsm[i] = 0.f;
    stage_layer0<H>(w0s, theta, nd, NTHR);
    __syncthreads();
    const float* Wo = w0s + (4 + L - 1) * H;
    for (int64_t tile = g; tile < ntiles; tile += G) {
        const int64_t pm = tile * 16 + m;
        const bool valid = pm < N;
        const int64_t pl = valid ? pm : N - 1;
        float ob[NOUT][C];
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int c = 0; c < C; ++c) ob[o][c] = valid ? Ob[((int64_t)c * NOUT + o) * ld + pm] : 0.f;
        f32x4 st[C];
        if (L - 1 >= 1) {
            const float* sp = stored + ((((size_t)tile * (L - 1) + (L - 2)) * C) * NT + ktile) * 256 + lane * 4;
#pragma unroll
            for (int c = 0; c < C; ++c) st[c] = *reinterpret_cast<const f32x4*>(sp + (size_t)c * NT * 256);
        } else {
            float xv[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; ++k) if (k < dim) xv[k] = pts_at(x, pl, dim, k);
            layer0_st<H, C, E>(w0s, xv, ktile, q, st);
        }
        f32x4 wo[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) wo[o] = *reinterpret_cast<const f32x4*>(&Wo[o * H + 16 * ktile + 4 * q]);
        float gwo[NOUT][4];
        f32x4 zb[C];
        {
            f32x4 a4[C], ab4[C];
            act_from_stored<D, E>(st[0], st + 1, st + 1 + D, shift, a4);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                f32x4 gg = (f32x4)(0.f);
#pragma unroll
                for (int c = 0; c < C; ++c) gg = gpe_fma((f32x4)(ob[o][c]), a4[c], gg);
#pragma unroll
                for (int r = 0; r < 4; ++r) gwo[o][r] = gg[r];
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
#pragma unroll
        for (int c = 0; c < C; ++c)
            *reinterpret_cast<f32x4*>(&Zout[(((size_t)tile * C + c) * NT + ktile) * 256 + lane * 4]) = zb[c];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) row_reduce4_add(gwo[o], &gsm[o * H + 16 * ktile], m, q);
        if (h == 0 && w == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float gbo = row_sum16(ob[o][0]);
                if (lane == 0) atomicAdd(&gsm[NOUT * H + o], gbo);
            }
        }
    }
    __syncthreads();
    float* slab = gslab + (size_t)g * Ppad;
    for (int i = threadIdx.x; i < NOUT * 16 * KTL; i += NTHR) {
        const int o = i / (16 * KTL), f = 16 * KTL * h + i % (16 * KTL);
        slab[nd.offW[L] + o * H + f] = gsm[o * H + f];
    }
    if (h == 0) for (int i = threadIdx.x; i < NOUT; i += NTHR) slab[nd.offB[L] + i] = gsm[NOUT * H + i];
}

#undef W_TRACE_K
#define W_TRACE_K 1
#define W_MAP_PIPE(H, C) ((H) == 128 && (C) <= 4)
#ifndef GPE_WIDE_ALT_PRIO
#define GPE_WIDE_ALT_PRIO 0   // w_bwd_map: the SIMD partners alternate s_setprio per K tile inside the product phases (as w_forward does)
#endif
#ifndef GPE_WIDE_STAGGER
#define GPE_WIDE_STAGGER 0    // w_bwd_map, one-barrier form: 1 = waves 4..7 run the deferred products BEFORE the adjoint phase (measured: 0.705 vs 0.735)
#endif
#ifndef GPE_WIDE_TOP_EARLY
#define GPE_WIDE_TOP_EARLY 0  // w_bwd_map, top launch, one-barrier form: the next tile's seeds / stored jets requested BEFORE the adjoint phase and its output-layer block
                              // (activation jets, W_out products, activation adjoint) placed in front of the wave's own weight-gradient products instead of behind them
#endif
#ifndef GPE_WIDE_SWP
#define GPE_WIDE_SWP 0        // w_bwd_map: LDS operand fragments of product group g+1 requested BEFORE the products of group g (1-step software pipeline)
#endif
#ifndef GPE_WIDE_PRIO_ACT
#define GPE_WIDE_PRIO_ACT 0   // w_bwd_map: > 0 raises the wave's priority over its activation block, < 0 over its product phases
#endif
// ---- reverse, one hidden->hidden map ------------------------------------------------------------------------------------------
// Zin = zbar_j, Zout = zbar_{j-1}: [tile][C][NT][256].  FIRST (j == 1): layer j-1 = 0 is recomputed from x, its gradients
// (dW_0, db_0) are formed here and nothing is written to Zout.
// LDS: gb[H] (db_j) | g0[4][H] | w0s | ZB[C][NT][256] (fragments at a swizzled slot) | XT[C][8][F_TILE]; H = 128 with C <= 4: two of each
// TOP = n_out (1 or 2) for the launch of the topmost map j = L-1, 0 otherwise: there zbar_j is not read from HBM but formed
// from the seeds Ob and the stored activations of the last hidden layer (what w_bwd_out does as a kernel of its own), and
// dW_out / db_out are accumulated here -- one launch and one HBM round trip of the adjoint jets fewer.
template <int H, int C, int E, int NSPLIT, bool FIRST, int TOP>
__global__ __launch_bounds__(512, 2) void w_bwd_map(NetDesc nd, int j, const float* __restrict__ theta,
                                                    const float* __restrict__ WpkT, Pts x,
                                                    const float* __restrict__ stored, const float* __restrict__ Zin,
                                                    float* __restrict__ Zout, float* __restrict__ gslab, int64_t N, int Ppad,
                                                    int G, const float* __restrict__ Ob, int64_t ld) {
    constexpr int D = C - 1 - E, NT = H / 16, NTHR = 64 * W_NW, RTZ = NT / W_NW, KTL = NT / NSPLIT;
    constexpr int NO = TOP > 0 ? TOP : 1;
    static_assert(KTL == W_NW, "one layer-(j-1) feature tile per wave");
    extern __shared__ __attribute__((aligned(16))) float lds_m[];
    float* gb = lds_m;
    float* g0 = gb + H;
    float* go = g0 + 4 * H;                                      // [2][H] dW_out | [4] db_out (TOP)
    float* w0s = go + 2 * H + 4;
    constexpr bool PIPE = W_MAP_PIPE(H, C);                       // both exchange buffers doubled, one barrier per tile (below)
    constexpr int ZSZ = C * NT * 256, XSZ = C * KTL * F_TILE;
    float* ZB = w0s + ((small_count(nd, H) + 3) & ~3);
    float* XT = ZB + (PIPE ? 2 : 1) * ZSZ;
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4, w = threadIdx.x >> 6;
    int h, g;
    w_block_role(blockIdx.x, G, NSPLIT, h, g);
    const int L = nd.n_lin - 1;
    const int dim = nd.dim;
    const float shift = nd.shift;
    const int64_t ntiles = (N + 15) >> 4;
    const int ktile = h * KTL + w;                               // this wave's feature tile of layer j-1
    // Every HBM / L2 access of the tile loop goes through a buffer descriptor: base + size in SGPRs, ONE per-lane byte offset (lane * 16)
    // for all of them, the per-(tile, channel, feature tile) part computed on the scalar unit -- plain pointer arithmetic costs two
    // 64-bit VALU adds per access here (~35 VALU instructions per tile and wave, and every VALU cycle is taken from the fp32 MFMAs)
    const int wu = __builtin_amdgcn_readfirstlane(w);            // (provably wave-uniform: the descriptor offsets stay in SGPRs)
    const int ktu = h * KTL + wu;
    const unsigned lane16 = (unsigned)lane * 16u;
    constexpr unsigned TILE_B = (unsigned)(C * NT * 256 * sizeof(float));     // one tile of adjoint jets / of one layer's stored jets
    for (int i = threadIdx.x; i < 7 * H + 4; i += NTHR) gb[i] = 0.f; // gb, g0 and go are contiguous
    if constexpr (FIRST || TOP > 0) stage_layer0<H>(w0s, theta, nd, NTHR);
    const float* Wo = w0s + (4 + L - 1) * H;
    const buf_t wbuf = buf_make(WpkT + (size_t)(j - 1) * H * H, (unsigned)(H * H * sizeof(float)));
    const buf_t obuf = buf_make(Ob, (unsigned)((size_t)C * NO * ld * sizeof(float)));      // TOP: the seeds [C][n_out][ld]
    int wofs = 0;                                               // opaque zero: keeps the weight loads inside the tile loop
    auto load_w = [&](int nt) { return buf_load4(wbuf, lane16, (unsigned)(wofs + (ktu * NT + nt) * 1024)); };
    f32x4 dwacc[RTZ][KTL];                                       // rows 16(w RTZ + rt).., column tiles of this half
#pragma unroll
    for (int rt = 0; rt < RTZ; ++rt)
#pragma unroll
        for (int kt = 0; kt < KTL; ++kt) dwacc[rt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbacc[RTZ];
#pragma unroll
    for (int rt = 0; rt < RTZ; ++rt) dbacc[rt] = 0.f;
    // H = 128 (register room): the small gradients of the launch -- db_j, and dW_out / db_out (TOP), dW_0 / db_0 (FIRST) -- are summed PER LANE
    // over all the tiles of the workgroup and reduced across the point lanes ONCE after the loop; per tile that was a DPP / bpermute
    // reduction plus an LDS atomic (~100 cycles of the wave each) for every one of them
    constexpr bool LANEACC = (H == 128);
    float gwoacc[NO][RTZ][4], gboacc[NO], g0acc[4][4];
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        gboacc[o] = 0.f;
#pragma unroll
        for (int rt = 0; rt < RTZ; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) gwoacc[o][rt][r] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) g0acc[k][r] = 0.f;
    __syncthreads();

    // The wave's rows of zbar_j (HBM, needed first thing in a tile) are requested one tile AHEAD, before the weight-gradient
    // products of the previous tile: their latency hides behind ~10 000 cycles of MFMAs instead of stalling all eight waves at
    // the top of every tile.  (The stored activations and the first weight chunk are needed only after the barriers / the
    // adjoint products, so they are requested at the top of their own tile.)
    f32x4 zf[RTZ][C];
    float xv[3] = {0.f, 0.f, 0.f};
    float obv[NO][C];                                            // TOP: the seeds of this lane's point
    auto issue_loads = [&](int64_t t) {
        if constexpr (TOP > 0) {                                 // stored (t, z_k, z_L) of layer j = L-1, own rows; converted at the tile top
            const buf_t sb = buf_make(stored + ((size_t)t * (L - 1) + (j - 1)) * (C * NT * 256), TILE_B);
#pragma unroll
            for (int rt = 0; rt < RTZ; ++rt)
#pragma unroll
                for (int c = 0; c < C; ++c) zf[rt][c] = buf_load4(sb, lane16, (unsigned)((c * NT + wu * RTZ + rt) * 1024));
            const int64_t pm = t * 16 + m;
            const unsigned pm4 = (unsigned)(pm * 4);             // (ld * C * NO * 4 bytes < 4 GB: checked by the launcher)
#pragma unroll
            for (int o = 0; o < NO; ++o)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float v = buf_load1(obuf, pm4, (unsigned)((c * NO + o) * (unsigned)ld * 4u));
                    obv[o][c] = pm < N ? v : 0.f;
                }
        } else {
            const buf_t zb_ = buf_make(Zin + (size_t)t * (C * NT * 256), TILE_B);
#pragma unroll
            for (int rt = 0; rt < RTZ; ++rt)
#pragma unroll
                for (int c = 0; c < C; ++c) zf[rt][c] = buf_load4(zb_, lane16, (unsigned)((c * NT + wu * RTZ + rt) * 1024));
        }
        if constexpr (FIRST) {
            const int64_t pm = t * 16 + m;
            const int64_t pl = pm < N ? pm : N - 1;
#pragma unroll
            for (int k = 0; k < 3; ++k) if (k < dim) xv[k] = pts_at(x, pl, dim, k);
        }
    };
    // z fragments live in ZB at a swizzled slot: the B-operand reads (every wave, all of ZB) and the TRANSPOSED reads of the wave's own
    // rows (A operands of the weight-gradient products: feature on lane) are both conflict-free, and no transposition scratch is needed
    const int zfrag = 4 * ((m ^ q) + 16 * q);
    int ztr[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) ztr[s2] = 4 * (((4 * q + s2) ^ (m >> 2)) + 16 * (m >> 2)) + (m & 3);
    WSTAMP_INIT;
    // zbar_j of a tile, own rows, into a z buffer.  TOP: formed here from the seeds and the stored activations of the last hidden layer
    // (dW_out / db_out on the way); otherwise it is what issue_loads fetched.
    auto publish_compute = [&]() {
        if constexpr (TOP > 0) {                                 // output layer: dW_out, db_out, zbar_{L-1} = act-adjoint(W_out^T Ob), own rows
#pragma unroll
            for (int rt = 0; rt < RTZ; ++rt) {
                f32x4 wo[NO];
#pragma unroll
                for (int o = 0; o < NO; ++o) wo[o] = *reinterpret_cast<const f32x4*>(&Wo[o * H + 16 * (w * RTZ + rt) + 4 * q]);
                float gwo[NO][4];
                {
                    f32x4 a4[C], ab4[C], zv4[C];
                    act_from_stored<D, E>(zf[rt][0], zf[rt] + 1, zf[rt] + 1 + D, shift, a4);
#pragma unroll
                    for (int o = 0; o < NO; ++o) {
                        f32x4 gg = (f32x4)(0.f);
#pragma unroll
                        for (int c = 0; c < C; ++c) gg = gpe_fma((f32x4)(obv[o][c]), a4[c], gg);
                        asm volatile("" : "+v"(gg));           // formed here, not sunk below the adjoint (a4 would stay live)
#pragma unroll
                        for (int r = 0; r < 4; ++r) gwo[o][r] = gg[r];
                    }
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        f32x4 v = (f32x4)(0.f);
#pragma unroll
                        for (int o = 0; o < NO; ++o) v = gpe_fma(wo[o], (f32x4)(obv[o][c]), v);
                        ab4[c] = v;
                    }
                    act_adjoint<D, E>(zf[rt][0], zf[rt] + 1, zf[rt] + 1 + D, ab4, zv4);   // zf holds the stored (t, z_k, z_L) here
#pragma unroll
                    for (int c = 0; c < C; ++c) zf[rt][c] = zv4[c];
                }
                if (h == 0) {
#pragma unroll
                    for (int o = 0; o < NO; ++o) {
                        if constexpr (LANEACC) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) gwoacc[o][rt][r] += gwo[o][r];
                        } else row_reduce4_add(gwo[o], &go[o * H + 16 * (w * RTZ + rt)], m, q);
                    }
                }
            }
            if (h == 0 && w == 0) {
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    if constexpr (LANEACC) gboacc[o] += obv[o][0];
                    else {
                        const float gbo = row_sum16(obv[o][0]);
                        if (lane == 0) atomicAdd(&go[2 * H + o], gbo);
                    }
                }
            }
        }
    };
    auto publish_store = [&](float* zbuf) {
#pragma unroll
        for (int rt = 0; rt < RTZ; ++rt)
#pragma unroll
            for (int c = 0; c < C; ++c) *reinterpret_cast<f32x4*>(&zbuf[(c * NT + w * RTZ + rt) * 256 + zfrag]) = zf[rt][c];
    };
    auto publish = [&](float* zbuf) { publish_compute(); publish_store(zbuf); };
    f32x4 st[C];
    f32x4 wn[W_KCB];
    float xk[3] = {0.f, 0.f, 0.f};
    auto load_st = [&](int64_t t) {
        if constexpr (!FIRST) {
            const buf_t sb = buf_make(stored + ((size_t)t * (L - 1) + (j - 2)) * (C * NT * 256), TILE_B);
#pragma unroll
            for (int c = 0; c < C; ++c) st[c] = buf_load4(sb, lane16, (unsigned)((c * NT + ktu) * 1024));
        }
#pragma unroll
        for (int i = 0; i < W_KCB; ++i) wn[i] = load_w(i);
    };
    // adjoint phase of a tile: abar (own feature tile of layer j-1) = sum_nt W_j^T[ktile, nt] zbar_j[nt] from `zbuf` (all rows), activation
    // adjoint -> zbar_{j-1} (HBM; FIRST: layer-0 gradients), X_{j-1} of the own feature tile -> `xbuf`
    auto adjoint_phase = [&](int64_t tile, const float* zbuf, float* xbuf) {
        f32x4 acc[C];                                            // C independent accumulator chains
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (GPE_WIDE_SWP) {
            // one K tile of lookahead on the B fragments: the compiler otherwise re-uses the fragment registers of a chunk for the next
            // one and can only request them when the last product of the chunk has issued -- every chunk then opens with an exposed
            // LDS round trip (~150 cycles per 32 products) unless the SIMD partner happens to have products ready
            f32x4 bfn[C];
#pragma unroll
            for (int c = 0; c < C; ++c) bfn[c] = *reinterpret_cast<const f32x4*>(&zbuf[(c * NT + 0) * 256 + zfrag]);
#pragma unroll
            for (int n0 = 0; n0 < NT; n0 += W_KCB) {
                f32x4 wv[W_KCB];
#pragma unroll
                for (int i = 0; i < W_KCB; ++i) wv[i] = wn[i];
                if (n0 + W_KCB < NT) {
#pragma unroll
                    for (int i = 0; i < W_KCB; ++i) wn[i] = load_w(n0 + W_KCB + i);
                }
#pragma unroll
                for (int i = 0; i < W_KCB; ++i) {
                    f32x4 bf[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) bf[c] = bfn[c];
                    if (n0 + i + 1 < NT) {
#pragma unroll
                        for (int c = 0; c < C; ++c) bfn[c] = *reinterpret_cast<const f32x4*>(&zbuf[(c * NT + n0 + i + 1) * 256 + zfrag]);
                    }
                    __builtin_amdgcn_sched_barrier(0);          // requests of the next K tile (and weight chunk) before this tile's products
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][s2], bf[c][s2], acc[c], 0, 0, 0);
                }
            }
        } else
#pragma unroll
        for (int n0 = 0; n0 < NT; n0 += W_KCB) {
            f32x4 wv[W_KCB];
#pragma unroll
            for (int i = 0; i < W_KCB; ++i) wv[i] = wn[i];
            if (n0 + W_KCB < NT) {
#pragma unroll
                for (int i = 0; i < W_KCB; ++i) wn[i] = load_w(n0 + W_KCB + i);
                __builtin_amdgcn_sched_barrier(0);              // issued before this chunk's products
            }
#pragma unroll
            for (int i = 0; i < W_KCB; ++i) {
                if constexpr (GPE_WIDE_ALT_PRIO) {               // SIMD partners (w, w + 4) take turns at the matrix pipe, one K tile each
                    if ((((n0 + i) & 1) == 0) == (w < W_NW / 2)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                }
                f32x4 bf[C];
#pragma unroll
                for (int c = 0; c < C; ++c) bf[c] = *reinterpret_cast<const f32x4*>(&zbuf[(c * NT + n0 + i) * 256 + zfrag]);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][s2], bf[c][s2], acc[c], 0, 0, 0);
            }
        }
        if constexpr (GPE_WIDE_ALT_PRIO) __builtin_amdgcn_s_setprio(0);
        WSTAMP(4);
        if constexpr (GPE_WIDE_PRIO_ACT != 0) __builtin_amdgcn_s_setprio(GPE_WIDE_PRIO_ACT > 0 ? GPE_WIDE_PRIO_ACT : 0);
        // recompute X of layer j-1 (own tile), activation adjoint -> zbar_{j-1}
        f32x4 xa[C], zb[C];
        act_from_stored<D, E>(st[0], st + 1, st + 1 + D, shift, xa);
        act_adjoint<D, E>(st[0], st + 1, st + 1 + D, acc, zb);
        if constexpr (!FIRST) {
            const buf_t ob_ = buf_make(Zout + (size_t)tile * (C * NT * 256), TILE_B);
#pragma unroll
            for (int c = 0; c < C; ++c) buf_store4(zb[c], ob_, lane16, (unsigned)((c * NT + ktu) * 1024));
        } else {      // linear map 0: g0[k][n] (k < dim: dW0[n][k]; k = 3: db0[n])
            const float z0[4] = {zb[0][0], zb[0][1], zb[0][2], zb[0][3]};
            if constexpr (LANEACC) {
#pragma unroll
                for (int r = 0; r < 4; ++r) g0acc[3][r] += z0[r];
            } else row_reduce4_add(z0, &g0[3 * H + 16 * ktile], m, q);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < D || (D == 0 && k < dim)) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = zb[0][r] * xk[k];
                        if constexpr (C > 1) { if (k < D) v[r] += zb[(1 + k) < C ? (1 + k) : 0][r]; }
                    }
                    if constexpr (LANEACC) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) g0acc[k][r] += v[r];
                    } else row_reduce4_add(v, &g0[k * H + 16 * ktile], m, q);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) xbuf[(c * KTL + w) * F_TILE + (4 * q + r) * F_PITCH + tr_wcol(m, q)] = xa[c][r];
        if constexpr (GPE_WIDE_PRIO_ACT != 0) __builtin_amdgcn_s_setprio(GPE_WIDE_PRIO_ACT > 0 ? 0 : -GPE_WIDE_PRIO_ACT);
    };
    // weight-gradient phase: dW_j[own rows][columns of this half] += Zbar^T X -- the wave's own rows of `zbuf` re-read feature-on-lane,
    // X_{j-1} (all feature tiles of this half) from `xbuf`: RTZ * 4 independent accumulator chains per chunk
    auto product_phase = [&](const float* zbuf, const float* xbuf) {
        f32x4 zt[RTZ][C];
#pragma unroll
        for (int rt = 0; rt < RTZ; ++rt) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float* zo = &zbuf[(c * NT + w * RTZ + rt) * 256];
                zt[rt][c] = (f32x4){zo[ztr[0]], zo[ztr[1]], zo[ztr[2]], zo[ztr[3]]};
            }
            if (h == 0) {                                        // bias gradient of map j: row sums of the value channel
                float s = (zt[rt][0][0] + zt[rt][0][1]) + (zt[rt][0][2] + zt[rt][0][3]);
                if constexpr (!LANEACC) {
                    s += __shfl_xor(s, 16, 64);
                    s += __shfl_xor(s, 32, 64);
                }
                dbacc[rt] += s;                                  // (LANEACC: the sum over the four point groups q follows after the loop)
            }
        }
        if constexpr (GPE_WIDE_SWP) {
            constexpr int NG = (KTL / 4) * C;                    // product groups: (column chunk of 4 tiles, channel)
            f32x4 xn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xn[i] = *reinterpret_cast<const f32x4*>(&xbuf[(0 * KTL + 0 + i) * F_TILE + tr_roff(m, q)]);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const int k0 = (gi / C) * 4, c = gi % C;
                f32x4 xf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[i] = xn[i];
                if (gi + 1 < NG) {
                    const int k1 = ((gi + 1) / C) * 4, c1 = (gi + 1) % C;
#pragma unroll
                    for (int i = 0; i < 4; ++i) xn[i] = *reinterpret_cast<const f32x4*>(&xbuf[(c1 * KTL + k1 + i) * F_TILE + tr_roff(m, q)]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int rt = 0; rt < RTZ; ++rt)
                            dwacc[rt][k0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[rt][c][s2], xf[i][s2], dwacc[rt][k0 + i], 0, 0, 0);
            }
        } else
#pragma unroll
        for (int k0 = 0; k0 < KTL; k0 += 4)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if constexpr (GPE_WIDE_ALT_PRIO) {
                    if ((((k0 / 4 * C + c) & 1) == 0) == (w < W_NW / 2)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                }
                f32x4 xf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    xf[i] = *reinterpret_cast<const f32x4*>(&xbuf[(c * KTL + k0 + i) * F_TILE + tr_roff(m, q)]);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int rt = 0; rt < RTZ; ++rt)
                            dwacc[rt][k0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[rt][c][s2], xf[i][s2], dwacc[rt][k0 + i], 0, 0, 0);
            }
        if constexpr (GPE_WIDE_ALT_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    if (g < ntiles) {
        issue_loads(g);
        if constexpr (PIPE) load_st(g);
        publish(ZB);
    }
    if constexpr (PIPE) {
        // H = 128, C <= 4: one workgroup per CU anyway, so both exchange buffers are doubled (137 KB) and the weight-gradient products of
        // a tile are DEFERRED behind the adjoint phase of the next one -- ONE barrier per tile, and between two barriers every wave has
        // adjoint products, activation arithmetic and weight-gradient products in a row: the two waves of a SIMD fall out of step by
        // themselves and one's VALU work runs under the other's matrix work.  (The first interval has no deferred products.)
        float *zr = ZB, *zw = ZB + ZSZ, *xw = XT, *xr = XT + XSZ;
        for (int64_t tile = g; tile < ntiles; tile += G) {
            WSTAMP_ITER;
            asm volatile("" : "+s"(wofs));
            if constexpr (FIRST) layer0_st<H, C, E>(w0s, xv, ktile, q, st);
#pragma unroll
            for (int k = 0; k < 3; ++k) xk[k] = xv[k];           // this tile's coordinates (xv is overwritten by the prefetch)
            WSTAMP(0);
            __syncthreads();                                     // zr (this tile's zbar_j) and xr (the previous tile's X) complete
            WSTAMP(1);
            const bool more = tile + G < ntiles;
            if (GPE_WIDE_STAGGER && wu >= W_NW / 2) {
                // the SIMD partners (waves w and w + 4) take the two product phases of the interval in opposite order: one's
                // activation arithmetic then runs under the other's matrix work instead of both reaching it together
                if (more) issue_loads(tile + G);
                __builtin_amdgcn_sched_barrier(0);
                if (tile != g) product_phase(zw, xr);
                WSTAMP(7);
                adjoint_phase(tile, zr, xw);
                if (more) load_st(tile + G);
                WSTAMP(5);
                if (more) publish(zw);
            } else if (TOP > 0 && GPE_WIDE_TOP_EARLY) {
                // top launch: the next tile's seeds and stored jets are requested a phase earlier, so that the output-layer block that turns them
                // into zbar_{L-1} can stand IN FRONT of the wave's weight-gradient products (the compiler is free to mix the two) instead of behind
                // them, where all eight waves ran it at the same time with the matrix pipe idle
                if (more) issue_loads(tile + G);
                adjoint_phase(tile, zr, xw);
                if (more) load_st(tile + G);
                __builtin_amdgcn_sched_barrier(0);
                WSTAMP(5);
                if (more) publish_compute();
                if (tile != g) product_phase(zw, xr);
                WSTAMP(7);
                if (more) publish_store(zw);
            } else {
                adjoint_phase(tile, zr, xw);
                if (more) { issue_loads(tile + G); load_st(tile + G); }  // in flight behind the products below
                __builtin_amdgcn_sched_barrier(0);
                WSTAMP(5);
                if (tile != g) product_phase(zw, xr);            // previous tile: own rows of zw, all of xr (the first interval has none)
                WSTAMP(7);
                if (more) publish(zw);                           // next tile's zbar_j over the rows just read
            }
            float* t0 = zr; zr = zw; zw = t0;
            t0 = xr; xr = xw; xw = t0;
        }
        __syncthreads();
        if (g < ntiles) product_phase(zw, xr);                   // the last tile's products
    } else {
        // Two barriers per tile.  After X: ZB holds zbar_j of the tile (all rows) and XT is free.  After Y: XT holds X_{j-1} of the tile and
        // nobody reads ZB any more -- so the weight-gradient products (own rows of ZB re-read transposed, XT) are followed by the NEXT
        // tile's rows going into ZB, and the barrier at the loop top closes both.
        for (int64_t tile = g; tile < ntiles; tile += G) {
            WSTAMP_ITER;
            asm volatile("" : "+s"(wofs));
            if constexpr (FIRST) layer0_st<H, C, E>(w0s, xv, ktile, q, st);
            load_st(tile);
#pragma unroll
            for (int k = 0; k < 3; ++k) xk[k] = xv[k];           // this tile's coordinates (xv is overwritten by the prefetch)
            WSTAMP(0);
            __syncthreads();                                     // X: ZB complete; the previous tile's products no longer read XT
            WSTAMP(1);
            adjoint_phase(tile, ZB, XT);
            const bool more = tile + G < ntiles;
            if (more) issue_loads(tile + G);                     // next tile's loads: in flight behind the products below
            __builtin_amdgcn_sched_barrier(0);
            WSTAMP(5);
            __syncthreads();                                     // Y: XT complete; ZB is no longer read by the adjoint products
            WSTAMP(6);
            product_phase(ZB, XT);
            WSTAMP(7);
            if (more) publish(ZB);                               // own rows of the next tile's zbar_j (the reads above came first)
        }
    }
    WSTAMP_FLUSH(8);
    // ---- slab: this workgroup's block of dW_j, db_j (h == 0), layer-0 gradients of its features (FIRST) ------------------------
    float* slab = gslab + (size_t)g * Ppad;
#pragma unroll
    for (int rt = 0; rt < RTZ; ++rt)
#pragma unroll
        for (int kt = 0; kt < KTL; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[nd.offW[j] + (16 * (w * RTZ + rt) + 4 * q + r) * H + 16 * (h * KTL + kt) + m] = dwacc[rt][kt][r];
    if constexpr (LANEACC) {
#pragma unroll
        for (int rt = 0; rt < RTZ; ++rt) {
            float sv = dbacc[rt];
            sv += __shfl_xor(sv, 16, 64);
            sv += __shfl_xor(sv, 32, 64);
            dbacc[rt] = sv;
        }
        if constexpr (TOP > 0) {
            if (g < ntiles && h == 0) {
#pragma unroll
                for (int o = 0; o < NO; ++o)
#pragma unroll
                    for (int rt = 0; rt < RTZ; ++rt) row_reduce4_add(gwoacc[o][rt], &go[o * H + 16 * (w * RTZ + rt)], m, q);
                if (w == 0) {
#pragma unroll
                    for (int o = 0; o < NO; ++o) {
                        const float gbo = row_sum16(gboacc[o]);
                        if (lane == 0) atomicAdd(&go[2 * H + o], gbo);
                    }
                }
            }
        }
        if constexpr (FIRST) {
            if (g < ntiles) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k == 3 || k < D || (D == 0 && k < dim)) row_reduce4_add(g0acc[k], &g0[k * H + 16 * ktile], m, q);
            }
        }
    }
    if (h == 0 && q == 0) {
#pragma unroll
        for (int rt = 0; rt < RTZ; ++rt) slab[nd.offB[j] + 16 * (w * RTZ + rt) + m] = dbacc[rt];
    }
    if constexpr (FIRST || TOP > 0) __syncthreads();
    if constexpr (FIRST) {
        for (int i = threadIdx.x; i < 4 * 16 * KTL; i += NTHR) {
            const int k = i / (16 * KTL), n = 16 * KTL * h + i % (16 * KTL);
            if (k == 3) slab[nd.offB[0] + n] = g0[3 * H + n];
            else if (k < dim) slab[nd.offW[0] + n * dim + k] = g0[k * H + n];
        }
    }
    if constexpr (TOP > 0) {
        if (h == 0) {
            for (int i = threadIdx.x; i < NO * H; i += NTHR) slab[nd.offW[L] + i] = go[i];
            for (int i = threadIdx.x; i < NO; i += NTHR) slab[nd.offB[L] + i] = go[2 * H + i];
        }
    }
}
