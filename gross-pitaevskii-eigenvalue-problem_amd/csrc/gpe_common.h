// gpe_common.h -- structs and device math shared by the generic and fused kernel sets (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gpe_hip.h"

#define GPE_DEV __device__ __forceinline__

// Network description passed by value to kernels.  Linear map j: width[j] -> width[j+1], j = 0..n_lin-1.
struct NetDesc {
    int n_lin;                       // number of nn.Linear
    int dim;                         // width[0]
    int n_out;                       // width[n_lin]
    float shift;                     // 0 (tanh) or 1 (tanh+1)  (plain MLP: every hidden layer; fused / wide kernel sets)
    float shiftv[GPE_MAX_LAYERS];    // per hidden layer (generic set; residual blocks use plain tanh behind a shifted first layer)
    int skip[GPE_MAX_LAYERS];        // map j: hidden layer whose activation jets are added to its output before the activation, or -1
    int width[GPE_MAX_LAYERS];
    int offW[GPE_MAX_LAYERS];        // offsets into the flat (torch-order) parameter vector
    int offB[GPE_MAX_LAYERS];
    int n_params;
};

// Physics of the head, passed by value.
struct Phys {
    int dim, n_out, complex_psi;
    float kin;                       // kinetic_coeff
    int potential;
    float pot_scale, omega[3], pot_a, pot_v0, pot_k;
    float omega_rot;
    float gamma;
    int p, abs_power;
    int base_mode, base_deriv, base_kind, envelope;
    float box_L, env_L;
    float perturb_scale, bc_nn_scale;
    float w_pde, w_bc, w_norm, w_sym, w_orth, sym_sign, w_riesz;
    int riesz_kind;
    float dx;
    double n_global;                 // N of the means
    float inv_world;                 // 1/world_size: scales the replicated boundary batch
    int n_orth;
    int lambda_kind;                 // GPE_LAMBDA_*
    float w_reg_f, reg_f_eps, w_reg_lam, reg_lam_eps;
};

// Indices into the double "sums" exchange buffer (all-reduced over ranks between phase 1 and 2).
enum { S_NUM = 0, S_DEN = 1, S_SYM = 2, S_ORTH0 = 3, /* ..S_ORTH0+3 */ S_RZ_K = 7, S_RZ_P = 8, S_RZ_I = 9, S_RZ_L = 10 /* <L_z> of complex psi */, S_COUNT = 12 };
// Local (replicated, never exchanged) double scalars.
enum { LS_BC_SE2 = 0, LS_BC_CNT = 1, LS_COUNT = 4 };
// Tail of the float gradient exchange buffer.
enum { GT_SUM_R2 = 0, GT_MSE_SE2 = 1, GT_COUNT = 4 };

// Optimiser / scheduler state living on the device (one struct; updated by k_update).
struct OptDev {
    double lr;            // lr used by the NEXT step
    double lr0;
    double best;          // plateau: best loss
    int num_bad;
    int nonfinite;        // sticky flag: a non-finite loss/grad was seen, updates were skipped
    long long step;       // optimiser steps taken
    int stopped;          // early stop fired (refine/harmonic_pinn_simulation.py:389-400)
    int es_count;         // steps since the best loss
    double es_best;
    long long stop_step;
    double b1p, b2p;      // beta1^step, beta2^step (running products: the Adam bias corrections without a pow() per step)
};

struct OptCfg {
    float beta1, beta2, eps, clip_norm;
    int sched;
    float T_0, T_mult, eta_min;
    float factor; int patience; float min_lr, threshold;
    float stop_tol; int stop_patience;
};

// Point coordinates of a batch: rows [0, na) come from `a`, rows [na, n) from `b` -- the collocation points and, appended
// behind them, the boundary points (the boundary batch rides in the collocation batch's launches; nothing is copied, both
// arrays stay the caller's).  b == NULL: a plain batch.
struct Pts { const float* a; const float* b; long long na; };
GPE_DEV float pts_at(const Pts& P, long long p, int dim, int k) { return p < P.na ? P.a[p * dim + k] : P.b[(p - P.na) * dim + k]; }

// ------------------------------------------------------------------------------------------------
// tanh = 1 - 2/(exp(2x)+1): five instructions (two of them quarter-rate), no branch on |x| -- on gfx950 the fp32 MFMAs and the
// VALU share the FMA lanes (no co-execution), so every VALU instruction of the activation is time taken from the matrix products.
// abs error <= 2.5e-7 over the whole line (fp32 round-off of the 1 - 2r form; tests/test_gpu_parity.py::test_tanh_accuracy_through_forward).
// Shared by both kernel sets so that they agree bit for bit on the activation.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
GPE_DEV float gpe_fma(float a, float b, float c) { return fmaf(a, b, c); }
GPE_DEV f32x4 gpe_fma(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }

GPE_DEV float gpe_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // exp(2x): 0 for very negative x (-> -1), +inf for large x (-> 1)
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);         // v_exp_f32, v_rcp_f32: 1 ulp each
}
GPE_DEV f32x4 gpe_tanh(f32x4 x) {                                      // the same per element, the multiplies and adds packed
    const f32x4 a = x * 2.8853900817779268f;
    f32x4 e, r;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(a[i]);
    const f32x4 d = e + 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_rcpf(d[i]);
    return gpe_fma((f32x4)(-2.0f), r, (f32x4)(1.0f));
}

// Jet channels.  A batch carries C = 1 + D + E channels per feature: the value, D first derivatives and E second-order
// channels.  E = D ("full"): the diagonal second derivatives d2/dx_k^2 one by one.  E = 1 with D > 1 ("Laplacian"): only their
// SUM -- every map of the network is linear in the second-order channels and the physics needs nothing but the Laplacian
// (k_head_pde), so the training batches propagate one summed channel: 4 channels instead of 5 in 2D, 5 instead of 7 in 3D.
//
// activation jets:  a = t + shift, a_k = s z_k, a_kk = s z_kk - 2 t s z_k^2   (t = tanh z, s = 1 - t^2)
//                   Laplacian channel: a_L = s z_L - 2 t s sum_k z_k^2
// With sigma = tanh:  sigma' = s = 1 - t^2,  sigma'' = w2 = -2 t s,  sigma''' = q = s (4 t^2 - 2 s) = s (4 - 6 s).
// The expressions below are arranged for the fewest VALU instructions (the compiler may not reassociate fp32): 10 for the
// jets, 26 for jets + adjoint of a 2D Laplacian batch.  Both kernel sets share them, so they agree bit for bit.
// T = float (one element) or f32x4 (the four elements a lane holds of one MFMA tile: packed v_pk_* arithmetic, the same
// operations in the same order per element, so both forms agree bit for bit).
template <int D, int E, typename T>
GPE_DEV void act_from_stored(T t, const T* zk, const T* zkk, float shift, T* a /*[1+D+E]*/) {
    const T s = gpe_fma(-t, t, (T)(1.0f));
    const T w2 = -2.0f * (t * s);
    a[0] = t + shift;
    if constexpr (E == D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            a[1 + j] = s * zk[j];
            a[1 + D + j] = gpe_fma(w2 * zk[j], zk[j], s * zkk[j]);
        }
    } else {
        static_assert(E == 1, "second-order channels: one per axis, or the single Laplacian channel");
        T S = zk[0] * zk[0];
#pragma unroll
        for (int j = 1; j < D; ++j) S = gpe_fma(zk[j], zk[j], S);
#pragma unroll
        for (int j = 0; j < D; ++j) a[1 + j] = s * zk[j];
        a[1 + D] = gpe_fma(w2, S, s * zkk[0]);
    }
}

// adjoint of the activation jets: given abar (adjoint of a-jets) and stored (t, z_k, z_kk) -> zbar
//   zbar_kk = s abar_kk
//   zbar_k  = s abar_k + 2 w2 z_k abar_kk
//   zbar    = s abar + sum_k [ w2 z_k abar_k + (w2 z_kk + q z_k^2) abar_kk ]
// Laplacian channel: the same with abar_kk -> abar_L for every k, z_kk -> z_L once, z_k^2 -> sum_k z_k^2.
template <int D, int E, typename T>
GPE_DEV void act_adjoint(T t, const T* zk, const T* zkk, const T* ab /*[1+D+E]*/, T* zb /*[1+D+E]*/) {
    const T s = gpe_fma(-t, t, (T)(1.0f));
    const T w2 = -2.0f * (t * s);
    const T q = s * gpe_fma((T)(-6.0f), s, (T)(4.0f));
    T acc = s * ab[0];
    if constexpr (E == D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const T akb = ab[1 + j], akkb = ab[1 + D + j];
            const T wz = w2 * zk[j];
            zb[1 + D + j] = s * akkb;
            zb[1 + j] = gpe_fma(wz + wz, akkb, s * akb);
            acc = gpe_fma(wz, akb, acc);
            acc = gpe_fma(gpe_fma(q * zk[j], zk[j], w2 * zkk[j]), akkb, acc);
        }
    } else {
        static_assert(E == 1, "second-order channels: one per axis, or the single Laplacian channel");
        const T aLb = ab[1 + D];
        zb[1 + D] = s * aLb;
        const T g = (w2 + w2) * aLb;
        T S = zk[0] * zk[0], dot = zk[0] * ab[1];
#pragma unroll
        for (int j = 1; j < D; ++j) { S = gpe_fma(zk[j], zk[j], S); dot = gpe_fma(zk[j], ab[1 + j], dot); }
#pragma unroll
        for (int j = 0; j < D; ++j) zb[1 + j] = gpe_fma(g, zk[j], s * ab[1 + j]);
        acc = gpe_fma(w2, dot, acc);
        acc = gpe_fma(gpe_fma(q, S, w2 * zkk[0]), aLb, acc);
    }
    zb[0] = acc;
}

// Riesz energy  E = (ak sum |grad u|^2 + ap sum V u^2 + ai sum |u|^(p+1)) / (normalised ? sum u^2 : 1)   (gpe_hip.h: riesz_kind)
GPE_DEV void riesz_coefs(const Phys& ph, float& ak, float& ap, float& ai, bool& normalised) {
    const float gi = ph.gamma / (float)(ph.p + 1);
    if (ph.riesz_kind == GPE_RIESZ_SUM) { ak = 0.5f; ap = 0.5f; ai = gi; normalised = false; }
    else if (ph.riesz_kind == GPE_RIESZ_VARIATIONAL) { ak = ph.kin; ap = 1.0f; ai = 2.0f * gi; normalised = true; }
    else { ak = 0.5f; ap = 1.0f; ai = gi; normalised = true; }
}

// The head kernel files the three energy sums with the Riesz coefficients on them; the energy-functional eigenvalue estimate
// (GPE_LAMBDA_ENERGY: src/gross_pitaevskii_2D.py:192) is their c / 1 / gamma combination over sum u^2.  gamma sum |u|^(p+1) is
// (p+1)/k times the interaction sum for ai = k gamma / (p+1), so gamma = 0 needs no division.
GPE_DEV bool phys_needs_energy_sums(const Phys& ph) { return ph.w_riesz != 0.f || ph.lambda_kind == GPE_LAMBDA_ENERGY; }
GPE_DEV double energy_numerator(const Phys& ph, const double* sums) {
    float ak, ap, ai; bool nrm;
    riesz_coefs(ph, ak, ap, ai, nrm);
    const double ci = ph.riesz_kind == GPE_RIESZ_VARIATIONAL ? 0.5 * (double)(ph.p + 1) : (double)(ph.p + 1);
    return (double)ph.kin * sums[S_RZ_K] / (double)ak + sums[S_RZ_P] / (double)ap + ci * sums[S_RZ_I];
}
GPE_DEV double lambda_of(const Phys& ph, const double* sums, double num, double den) {
    return ph.lambda_kind == GPE_LAMBDA_ENERGY ? energy_numerator(ph, sums) / den : num / den;
}
// w_reg_f / (mean u^2 + eps_f) + w_reg_lam / (lambda^2 + eps_l)    (src/gross_pitaevskii_2D.py:197-211)
GPE_DEV double reg_terms(const Phys& ph, double den, double lam) {
    double r = 0.0;
    if (ph.w_reg_f != 0.f) r += (double)ph.w_reg_f / (den / ph.n_global + (double)ph.reg_f_eps);
    if (ph.w_reg_lam != 0.f) r += (double)ph.w_reg_lam / (lam * lam + (double)ph.reg_lam_eps);
    return r;
}

GPE_DEV float ipowf(float u, int p) {
    float r = 1.0f;
    for (int i = 0; i < p; ++i) r *= u;
    return r;
}

// phi_n(x), phi_n', phi_n'' of the 1D harmonic oscillator (refine/harmonic_pinn_simulation.py:95-119).
GPE_DEV void hermite_base(float x, int n, int deriv_mode, float norm, float& phi, float& phi1, float& phi2) {
    float Hm2 = 0.f, Hm1 = 0.f, H = 1.f;
    for (int k = 0; k < n; ++k) {
        float Hn = 2.f * x * H - 2.f * (float)k * Hm1;
        Hm2 = Hm1; Hm1 = H; H = Hn;
    }
    float w = expf(-0.5f * x * x);
    float H1 = 0.f, H2 = 0.f;
    if (deriv_mode == 0) {
        H1 = 2.f * (float)n * Hm1;
        H2 = 4.f * (float)n * (float)(n - 1) * Hm2;
    }
    phi = norm * (H * w);
    phi1 = norm * w * (H1 - x * H);
    phi2 = norm * w * (H2 - 2.f * x * H1 + (x * x - 1.f) * H);
}

// base phi_n and its first two derivatives at a 1D point (kinds HERMITE / BOX)
GPE_DEV void base_at(const Phys& ph, float x, float hermite_norm, float& phi, float& phi1, float& phi2) {
    if (ph.base_kind == GPE_BASE_BOX) {               // refine/box_pinn_simulation.py:99-117,141-180
        const float k = (float)(ph.base_mode + 1) * 3.14159265358979323846f / ph.box_L;
        const float a = sqrtf(2.0f / ph.box_L);
        float sn, cs;
        sincosf(k * x, &sn, &cs);
        phi = a * sn; phi1 = a * k * cs; phi2 = -a * k * k * sn;
    } else {
        hermite_base(x, ph.base_mode, ph.base_deriv, hermite_norm, phi, phi1, phi2);
    }
}

// hard boundary factor f(x) = sin(pi x / L) and derivatives (refine/box_pinn_simulation.py:127-130)
GPE_DEV void envelope_at(const Phys& ph, float x, float& f, float& f1, float& f2) {
    const float k = 3.14159265358979323846f / ph.env_L;
    float sn, cs;
    sincosf(k * x, &sn, &cs);
    f = sn; f1 = k * cs; f2 = -k * k * sn;
}

GPE_DEV float potential_at(const Phys& ph, const float* xv, const float* Vpre, int64_t m) {
    switch (ph.potential) {
        case GPE_POT_PRECOMPUTED: return Vpre[m];
        case GPE_POT_HARMONIC: {
            float V = 0.f;
            for (int k = 0; k < ph.dim; ++k) { float t = ph.omega[k] * (xv[k] - (k == 0 ? ph.pot_a : 0.f)); V = fmaf(t, t, V); }
            return ph.pot_scale * V;
        }
        case GPE_POT_GAUSSIAN: { float t = xv[0] - ph.pot_a; return expf(-t * t); }
        case GPE_POT_PERIODIC: { float c = cosf(ph.pot_k * xv[0]); return ph.pot_v0 * c * c; }
        default: return 0.f;
    }
}

// Seeds dLoss/d(output jets) of ONE collocation point for real psi without orthogonality / Riesz terms -- what k_seed_pde
// (gpe_head.h) writes to Ob for such a point, as a function: the reverse kernel of small batches forms them itself (SeedArgs, below)
// instead of reading them back, one launch fewer per step.  Returns r^2 of the point.  Same operations in the same order as the kernel.
template <int C, int E>
GPE_DEV float seed_point(const Phys& ph, const float* xv, float V, float u, float Hu, float lam, float I, float (&ob)[C]) {
    constexpr int D = C - 1 - E;
    const float r = Hu - lam * u;
    const float cr = (float)(2.0 * (double)ph.w_pde / ph.n_global);
    const float rb = cr * r;
    const float dint = ph.abs_power ? ph.gamma * (float)ph.p * ipowf(fabsf(u), ph.p - 1) : ph.gamma * (float)ph.p * ipowf(u, ph.p - 1);
    float ub = rb * (V + dint - lam);
    const float cn = ph.w_norm * 4.0f * (I - 1.0f) * ph.dx;
    ub += cn * u;
    float Ub[C];
    Ub[0] = ub;
#pragma unroll
    for (int j = 0; j < D; ++j) Ub[1 + j] = 0.f;
#pragma unroll
    for (int j = 0; j < E; ++j) Ub[1 + D + j] = -ph.kin * rb;
    if constexpr (D == 1) {
        if (ph.envelope == GPE_ENV_SIN) {      // adjoint of psi = o f
            float f, f1, f2;
            envelope_at(ph, xv[0], f, f1, f2);
            const float u0 = Ub[0], u1 = Ub[1], u2 = Ub[2];
            Ub[0] = fmaf(f, u0, fmaf(f1, u1, f2 * u2));
            Ub[1] = fmaf(f, u1, 2.0f * f1 * u2);
            Ub[2] = f * u2;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) ob[c] = ph.perturb_scale * Ub[c];
    return r * r;
}
// what a reverse kernel needs to form the seeds of the collocation rows itself (boundary rows riding in the batch keep the seeds
// the head kernel wrote to Ob)
struct SeedArgs {
    Phys ph;
    const float* Vpre; const float* u; const float* Hu;     // [ld] each (n_out = 1)
    const double* sums;                                      // S_NUM, S_DEN of this step (complete: the head kernel has finished)
    double* sum_r2;                                          // += sum of r^2 over the collocation rows
    int64_t n_pde;                                           // rows [0, n_pde) are collocation points
    // the forward kernel ran the head itself (HeadArgs, gpe_head.h) and left one (num, den, bse) triple per workgroup: every workgroup
    // of the reverse kernel adds the triples in index order (deterministic), workgroup 0 also files the totals for the update kernel
    const double* slots; int nslots;
    double* sums_out; double* lsums_out;
    // f_backward_pipe, every variant: > 0 = the tiles of a CU's workgroup pair are split unevenly (share of the first-dispatched one, / 1024)
    int old_share_q10;
};

// block-wide sum of a double over 256 threads -> valid in thread 0
GPE_DEV double block_sum_256(double v, double* red /* >= one double of LDS per wave of the block */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += red[i];
    }
    return r;
}
