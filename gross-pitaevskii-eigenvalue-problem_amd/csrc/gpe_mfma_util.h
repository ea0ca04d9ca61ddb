// gpe_mfma_util.h -- device helpers shared by the fused (gpe_fused.h) and wide (gpe_wide.h) kernel sets: fragment types, DPP row
// reductions, 16x16 tile transposes through LDS, layer-0 staging.  gfx950 only.
#pragma once
#include "gpe_common.h"



// Buffer addressing: descriptor (base, size) in SGPRs + one 32-bit per-lane byte offset + a wave-uniform byte offset that the
// scalar unit computes.  With plain pointers hipcc forms 64-bit per-lane addresses with VALU adds -- and on gfx950 every VALU
// cycle is a cycle the fp32 MFMAs do not get.  Out-of-range accesses are dropped by the hardware: sizes must be exact.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t buf_t;
GPE_DEV buf_t buf_make(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
GPE_DEV f32x4 buf_load4(buf_t r, unsigned lane_bytes, unsigned uni_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, uni_bytes, 0));
}
GPE_DEV float buf_load1(buf_t r, unsigned lane_bytes, unsigned uni_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_bytes, uni_bytes, 0));
}
// HAZARD (found the hard way, round 4: value-only batches with the shifted tanh): a 16-byte buffer store whose uniform offset sits in an
// SGPR, followed IMMEDIATELY by a VALU instruction that overwrites its data registers (the compiler forms a = t + shift in place right
// behind the store of t), stores the NEW values on gfx950.  hipcc's hazard recogniser pads this pattern only for stores WITHOUT a register
// soffset.  The empty-bodied wait below names the stored registers as inputs, so nothing can overwrite them before two wait states
// have passed (2 cycles per store; the stores of the hot loops are 1 in ~16 instructions).
GPE_DEV void buf_store4(f32x4 v, buf_t r, unsigned lane_bytes, unsigned uni_bytes) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, lane_bytes, uni_bytes, 0);
    asm volatile("s_nop 1" ::"v"(v));
}

// Transposition tiles: 16 x 16 floats, row pitch 16, the column index XOR-swizzled with 8 in rows 8..15.  Written point-on-lane
// (row 4q+r, column m: ds_write_b32, 2-way on the 32-bank write path -- free), read feature-on-lane (row m, columns 4q..4q+3: one
// ds_read_b128).  The 16-lane groups of ds_read_b128 are NOT contiguous on gfx950 ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS):
// with the former padded pitch of 20 every group had a 2-way conflict (8 instead of 4 LDS cycles per read, SQ_LDS_BANK_CONFLICT =
// 24 % of the reverse kernel's LDS cycles); with the swizzle every group covers the 16 slots of the bank row exactly once.
#define F_PITCH 16
GPE_DEV int tr_wcol(int m, int q) { return m ^ ((q & 2) << 2); }                        // column of point m in rows 4q..4q+3
GPE_DEV int tr_roff(int m, int q) { return m * F_PITCH + 4 * (q ^ ((m >> 3) << 1)); }   // float offset of (row m, columns 4q..4q+3)

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

// 16 values per lane, 16 lanes per DPP row: lane m of every row returns sum_{lanes of the row} v[m] -- a butterfly that
// halves the live values each step (row_mirror, row_half_mirror, quad xor 2, quad xor 1): 30 DPP adds + 15 selects for what
// 16 row_sum16 calls did in 64 DPP adds, and the 16 sums land on 16 different lanes, so ONE LDS atomic instruction (64 distinct
// addresses) replaces 16 four-lane ones (ds_add_f32 costs ~100 cycles of the wave whatever the lane count).
GPE_DEV float row_reduce_pick16(const float (&v)[16], int m) {
    const bool b3 = (m & 8) != 0, b2 = (m & 4) != 0, b1 = (m & 2) != 0, b0 = (m & 1) != 0;
    float a8[8], a4[4], a2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float lo = v[i] + dpp_mov<0x140>(v[i]), hi = v[i + 8] + dpp_mov<0x140>(v[i + 8]);
        a8[i] = b3 ? hi : lo;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = a8[i] + dpp_mov<0x141>(a8[i]), hi = a8[i + 4] + dpp_mov<0x141>(a8[i + 4]);
        a4[i] = b2 ? hi : lo;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float lo = a4[i] + dpp_mov<0x4E>(a4[i]), hi = a4[i + 2] + dpp_mov<0x4E>(a4[i + 2]);
        a2[i] = b1 ? hi : lo;
    }
    const float lo = a2[0] + dpp_mov<0xB1>(a2[0]), hi = a2[1] + dpp_mov<0xB1>(a2[1]);
    return b0 ? hi : lo;
}
// NF (= H/4) per-lane values, value f belonging to feature 16(f>>2) + 4q + (f&3): add their sums over the tile's 16 points to
// dst[feature * stride]; chunks of 16 values, one atomic instruction per chunk.
template <int NF>
GPE_DEV void row_reduce_add(const float (&v)[NF], float* dst, int stride, int m, int q) {
#pragma unroll
    for (int f0 = 0; f0 < NF; f0 += 16) {
        float c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = (f0 + i < NF) ? v[(f0 + i < NF) ? f0 + i : 0] : 0.f;
        const float t = row_reduce_pick16(c, m);
        const int fi = f0 + m;
        if (fi < NF) atomicAdd(&dst[(16 * (fi >> 2) + 4 * q + (fi & 3)) * stride], t);
    }
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
        for (int r = 0; r < 4; ++r) T[c * F_TILE + (4 * q + r) * F_PITCH + tr_wcol(m, q)] = v[c][r];
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = *reinterpret_cast<const f32x4*>(&T[c * F_TILE + tr_roff(m, q)]);
}

// ---- layer 0 helpers ---------------------------------------------------------------------------------------------
// W0 ([H][dim], dim <= 3) and b0 are staged once per workgroup into LDS as w0s[4][H]: rows 0..2 = W0^T zero-padded to
// three coordinates, row 3 = b0.  With the point coordinates zero-padded too, layer 0 is branch-free for any dim.
// Behind it, in the same LDS block: the hidden biases b_1..b_{L-1} ([L-1][H]), the output weights ([n_out][H]) and the
// output bias -- every small operand the per-tile code reads, so that no global (L2-latency) load sits inside a tile.
GPE_DEV int small_count(const NetDesc& nd, int H) { return (4 + (nd.n_lin - 2) + nd.n_out) * H + 4; }
template <int H>
GPE_DEV void stage_layer0(float* w0s, const float* __restrict__ theta, const NetDesc& nd, int nthr, int tix = -1) {
    const int L = nd.n_lin - 1;
    if (tix < 0) tix = threadIdx.x;               // (tix: thread index within the group of nthr threads that fills this copy)
    // The first pass of all four copy loops is requested TOGETHER, then written (round 4): as four loops in a row, each with its
    // offset-table load -> parameter load -> s_waitcnt vmcnt(0) -> ds_write, the prologue of every fused kernel was a chain of
    // eight L2 round trips -- 3-4 us before a workgroup's first tile, a tenth of a kernel at the reference's batch sizes.
    const int n1 = 4 * H, n2 = (L - 1) * H, n3 = nd.n_out * H, n4 = nd.n_out;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (tix < n1) {
        const int k = tix / H, n = tix % H;
        if (k == 3) v1 = theta[nd.offB[0] + n];
        else if (k < nd.dim) v1 = theta[nd.offW[0] + n * nd.dim + k];
    }
    if (tix < n2) v2 = theta[nd.offB[1 + tix / H] + tix % H];
    if (tix < n3) v3 = theta[nd.offW[L] + tix];
    if (tix < n4) v4 = theta[nd.offB[L] + tix];
    if (tix < n1) w0s[tix] = v1;
    if (tix < n2) w0s[4 * H + tix] = v2;
    if (tix < n3) w0s[(4 + L - 1) * H + tix] = v3;
    if (tix < n4) w0s[(4 + L - 1 + nd.n_out) * H + tix] = v4;
    for (int i = tix + nthr; i < n1; i += nthr) {
        const int k = i / H, n = i % H;
        float v;
        if (k == 3) v = theta[nd.offB[0] + n];
        else v = (k < nd.dim) ? theta[nd.offW[0] + n * nd.dim + k] : 0.f;
        w0s[i] = v;
    }
    for (int i = tix + nthr; i < n2; i += nthr) w0s[4 * H + i] = theta[nd.offB[1 + i / H] + i % H];
    for (int i = tix + nthr; i < n3; i += nthr) w0s[(4 + L - 1) * H + i] = theta[nd.offW[L] + i];
    for (int i = tix + nthr; i < n4; i += nthr) w0s[(4 + L - 1 + nd.n_out) * H + i] = theta[nd.offB[L] + i];
}

// global -> LDS copy of n16 16-byte elements by nthr threads: four loads per thread requested at once and the next four behind them
// before the first four are written (round 4: the plain loop compiled to ONE dwordx4 load -> s_waitcnt vmcnt(0) -> ds_write_b128 per
// pass, twelve L2 round trips in a row for three 64 x 64 maps).
template <typename V4>
GPE_DEV void stage_copy16(V4* __restrict__ dst, const V4* __restrict__ src, int n16, int tix, int nthr) {
    int i = tix;
    V4 c0, c1, c2, c3;
    const bool have = i + 3 * nthr < n16;
    if (have) { c0 = src[i]; c1 = src[i + nthr]; c2 = src[i + 2 * nthr]; c3 = src[i + 3 * nthr]; }
    while (i + 3 * nthr < n16) {
        const int j = i + 4 * nthr;
        const bool more = j + 3 * nthr < n16;
        V4 d0 = c0, d1 = c1, d2 = c2, d3 = c3;
        if (more) { d0 = src[j]; d1 = src[j + nthr]; d2 = src[j + 2 * nthr]; d3 = src[j + 3 * nthr]; }
        dst[i] = c0; dst[i + nthr] = c1; dst[i + 2 * nthr] = c2; dst[i + 3 * nthr] = c3;
        c0 = d0; c1 = d1; c2 = d2; c3 = d3;
        i = j;
    }
    for (; i < n16; i += nthr) dst[i] = src[i];
}

// stored-equivalent (t, z_k, z_kk) of hidden layer 0 for features 16nt+4q+r, recomputed from the point coordinates
template <int H, int C, int E>
GPE_DEV void layer0_st(const float* w0s, const float (&xv)[3], int nt, int q, f32x4 (&st)[C]) {
    constexpr int D = C - 1 - E;
    const int o = 16 * nt + 4 * q;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(&w0s[o]);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(&w0s[H + o]);
    const f32x4 w2 = *reinterpret_cast<const f32x4*>(&w0s[2 * H + o]);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(&w0s[3 * H + o]);
    const f32x4 z = gpe_fma(w2, (f32x4)(xv[2]), gpe_fma(w1, (f32x4)(xv[1]), gpe_fma(w0, (f32x4)(xv[0]), bb)));
    st[0] = gpe_tanh(z);
    if constexpr (D >= 1) st[1] = w0;
    if constexpr (D >= 2) st[2] = w1;
    if constexpr (D >= 3) st[3] = w2;
#pragma unroll
    for (int e = 0; e < E; ++e) st[1 + D + e] = (f32x4)(0.f);        // a linear map has no second derivatives
}

GPE_DEV void row_reduce4_add(const float (&v)[4], float* dst16, int m, int q) {
    // sum over the 16 point lanes of 4 per-lane values (features 4q+r of the slice); total of value r lands on lanes m>>2 == r
    const bool b3 = (m & 8) != 0, b2 = (m & 4) != 0;
    float a2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float lo = v[i] + dpp_mov<0x140>(v[i]), hi = v[i + 2] + dpp_mov<0x140>(v[i + 2]);
        a2[i] = b3 ? hi : lo;
    }
    const float lo = a2[0] + dpp_mov<0x141>(a2[0]), hi = a2[1] + dpp_mov<0x141>(a2[1]);
    float t = b2 ? hi : lo;
    t += dpp_mov<0x4E>(t);
    t += dpp_mov<0xB1>(t);
    if ((m & 3) == 0) atomicAdd(&dst16[4 * q + (m >> 2)], t);
}
