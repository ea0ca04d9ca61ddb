// gpe_head.h -- physics kernels shared by both MLP kernel sets: NN output jets -> (u, Hu, sums),
// sums -> lambda, residual, seeds (adjoint of the NN output jets); boundary and symmetry terms.
//
// Layout of the output jets O and their adjoint Ob: [C][n_out][ld] (point index contiguous).
// Reference: pde_loss refine/harmonic_pinn_simulation.py:146-196 (nb c6:L81-127), boundary_loss :198-210,
// normalization_loss :212-217, symmetry_loss nb c6:L137-155; 2D Laplacian template src/gross_pitaevskii_2D.py:183-195.
#pragma once
#include "gpe_common.h"

// u-jets (value, first, second derivatives) of component o at point m from NN output jets.
// bptr: device array of pointers; [0..3] orthogonality modes, [4..6] precomputed base phi, phi', phi''.
// (U holds the raw NN output jets of the component on entry)
template <int C, int E>
GPE_DEV void u_jets_from(const Phys& ph, int64_t m, const float* xv, float base_norm, const float* const* __restrict__ bptr, float* U /*[C]*/) {
#pragma unroll
    for (int c = 0; c < C; ++c) U[c] = ph.perturb_scale * U[c];
    if constexpr (C == 3 && E == 1) {
        if (ph.envelope == GPE_ENV_SIN) {              // psi = o f : product rule on the jets
            float f, f1, f2;
            envelope_at(ph, xv[0], f, f1, f2);
            const float o0 = U[0], o1 = U[1], o2 = U[2];
            U[0] = o0 * f;
            U[1] = fmaf(o1, f, o0 * f1);
            U[2] = fmaf(o2, f, fmaf(2.0f * o1, f1, o0 * f2));
        }
        if (ph.base_mode >= 0) {
            float phi, p1, p2;
            if (ph.base_kind == GPE_BASE_PRECOMPUTED) { phi = bptr[4][m]; p1 = bptr[5][m]; p2 = bptr[6][m]; }
            else base_at(ph, xv[0], base_norm, phi, p1, p2);
            U[0] += phi; U[1] += p1; U[2] += p2;
        }
    }
}
template <int C, int E>
GPE_DEV void load_u_jets(const Phys& ph, const float* __restrict__ O, int64_t ld, int64_t m, int o,
                         const float* xv, float base_norm, const float* const* __restrict__ bptr, float* U /*[C]*/) {
#pragma unroll
    for (int c = 0; c < C; ++c) U[c] = O[((int64_t)c * ph.n_out + o) * ld + m];
    u_jets_from<C, E>(ph, m, xv, base_norm, bptr, U);
}

// One row of a training batch of REAL psi without orthogonality / Riesz terms, from the raw NN output jets Oj[C] of the point -- what
// k_head_pde does for such a row, as a function: the cooperative forward kernel of small batches runs it in its epilogue (HeadArgs)
// and the head kernel is not launched.  Collocation row: u, H u to global memory, num += u Hu, den += u^2.  Boundary row riding in the
// batch (m >= n_pde): e = base + s NN - target, bse += e^2, seeds straight to Ob.
struct HeadArgs {
    Phys ph; float base_norm;
    const float* Vpre; const float* const* bptr; const float* bc_target;
    float* u; float* Hu; float* Ob;                          // [ld] / [C][ld]
    int64_t n_pde, ld;
    double* slots;                                           // [gridDim.x][4]: this workgroup's (num, den, bse) partial sums
};
template <int C, int E>
GPE_DEV void head_point_real(const HeadArgs& ha, const float* xv, int64_t m, int64_t N, const float* Oj, double& num, double& den, double& bse) {
    constexpr int D = C - 1 - E;
    const Phys& ph = ha.ph;
    if (m >= ha.n_pde) {
        const int64_t mb = m - ha.n_pde;
        const float cnt = (float)((N - ha.n_pde) * ph.n_out);
        float fenv = 1.0f;
        if (ph.envelope == GPE_ENV_SIN) { float f1, f2; envelope_at(ph, xv[0], fenv, f1, f2); }
        float e = ph.bc_nn_scale * fenv * Oj[0];
        if (ph.base_mode >= 0 && ph.base_kind != GPE_BASE_PRECOMPUTED) {
            float phi, p1, p2;
            base_at(ph, xv[0], ha.base_norm, phi, p1, p2);
            e += phi;
        }
        if (ha.bc_target) e -= ha.bc_target[mb * ph.n_out];
        bse += (double)(e * e);
        ha.Ob[m] = ph.w_bc * 2.0f / cnt * e * ph.bc_nn_scale * fenv * ph.inv_world;
#pragma unroll
        for (int c = 1; c < C; ++c) ha.Ob[(int64_t)c * ha.ld + m] = 0.f;
        return;
    }
    const float V = potential_at(ph, xv, ha.Vpre, m);
    float U[C];
#pragma unroll
    for (int c = 0; c < C; ++c) U[c] = Oj[c];
    u_jets_from<C, E>(ph, m, xv, ha.base_norm, ha.bptr, U);
    const float u = U[0];
    float lap = 0.f;
#pragma unroll
    for (int j = 0; j < E; ++j) lap += U[1 + D + j];
    const float inter = ph.abs_power ? ph.gamma * ipowf(fabsf(u), ph.p - 1) * u : ph.gamma * ipowf(u, ph.p);
    const float Hu = -ph.kin * lap + V * u + inter;
    ha.u[m] = u;
    ha.Hu[m] = Hu;
    num += (double)(u * Hu);
    den += (double)(u * u);
}

// ---- phase 1: u, Hu per point; block partial sums into sums[] (double atomics) ---------------------
template <int C, int E>
// Rows [n_pde, N) of a merged batch are boundary points (refine/harmonic_pinn_simulation.py:198-210): for those the kernel
// forms e = base + s*NN - target, adds e^2 to lsums[LS_BC_SE2] and writes their seeds Ob = w_bc*2/cnt * e * s / world directly
// (they do not depend on mu) -- what k_head_seed_bc does for a separate boundary batch.
__global__ __launch_bounds__(1024) void k_head_pde(Phys ph, float base_norm, Pts x,
                                                  const float* __restrict__ Vpre, const float* __restrict__ O,
                                                  const float* const* __restrict__ orth, float* __restrict__ u_out,
                                                  float* __restrict__ Hu_out, float* __restrict__ ux_out,
                                                  double* __restrict__ sums, int64_t N, int64_t ld, int64_t n_pde,
                                                  const float* __restrict__ bc_target, float* __restrict__ Ob,
                                                  double* __restrict__ lsums) {
    constexpr int D = C - 1 - E;
    double num = 0.0, den = 0.0, so[GPE_MAX_ORTH] = {0.0, 0.0, 0.0, 0.0};
    double rzk = 0.0, rzp = 0.0, rzi = 0.0, rzl = 0.0, bse = 0.0;
    // grid-stride: few workgroups, one double atomic each per sum (same-address atomics serialise at ~25 ns apiece)
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < N; m += (int64_t)gridDim.x * blockDim.x) {
        float xv[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < ph.dim; ++k) xv[k] = pts_at(x, m, ph.dim, k);
        if (m >= n_pde) {                              // boundary point riding in this batch
            const int64_t mb = m - n_pde;
            const float cnt = (float)((N - n_pde) * ph.n_out);
            float fenv = 1.0f;
            if (ph.envelope == GPE_ENV_SIN) { float f1, f2; envelope_at(ph, xv[0], fenv, f1, f2); }
            for (int o = 0; o < ph.n_out; ++o) {
                float e = ph.bc_nn_scale * fenv * O[(int64_t)o * ld + m];
                if (ph.base_mode >= 0 && o == 0 && ph.base_kind != GPE_BASE_PRECOMPUTED) {
                    float phi, p1, p2;
                    base_at(ph, xv[0], base_norm, phi, p1, p2);
                    e += phi;
                }
                if (bc_target) e -= bc_target[mb * ph.n_out + o];
                bse += (double)(e * e);
                Ob[(int64_t)o * ld + m] = ph.w_bc * 2.0f / cnt * e * ph.bc_nn_scale * fenv * ph.inv_world;
#pragma unroll
                for (int c = 1; c < C; ++c) Ob[((int64_t)c * ph.n_out + o) * ld + m] = 0.f;
            }
            continue;
        }
        float V = potential_at(ph, xv, Vpre, m);
        float U[2][C];
        for (int o = 0; o < ph.n_out; ++o) load_u_jets<C, E>(ph, O, ld, m, o, xv, base_norm, orth, U[o]);
        float Hu[2] = {0.f, 0.f};
        if (!ph.complex_psi) {
            float u = U[0][0], lap = 0.f;
#pragma unroll
            for (int j = 0; j < E; ++j) lap += U[0][1 + D + j];
            float inter = ph.abs_power ? ph.gamma * ipowf(fabsf(u), ph.p - 1) * u : ph.gamma * ipowf(u, ph.p);
            Hu[0] = -ph.kin * lap + V * u + inter;
        } else {
            float ur = U[0][0], ui = U[1][0];
            float rho = ur * ur + ui * ui;
            float lr = 0.f, li = 0.f;
#pragma unroll
            for (int j = 0; j < E; ++j) { lr += U[0][1 + D + j]; li += U[1][1 + D + j]; }
            Hu[0] = -ph.kin * lr + V * ur + ph.gamma * rho * ur;
            Hu[1] = -ph.kin * li + V * ui + ph.gamma * rho * ui;
            if (ph.omega_rot != 0.f && D >= 2) {
                float Dr = xv[0] * U[0][2] - xv[1] * U[0][1];
                float Di = xv[0] * U[1][2] - xv[1] * U[1][1];
                Hu[0] += -ph.omega_rot * Di;
                Hu[1] += ph.omega_rot * Dr;
            }
        }
        for (int o = 0; o < ph.n_out; ++o) {
            float u = U[o][0];
            u_out[(int64_t)o * ld + m] = u;
            Hu_out[(int64_t)o * ld + m] = Hu[o];
            num += (double)(u * Hu[o]);
            den += (double)(u * u);
        }
        for (int j = 0; j < ph.n_orth; ++j) so[j] += (double)(orth[j][m] * U[0][0]);
        if constexpr (D >= 1) {
            if (phys_needs_energy_sums(ph)) {   // Paper nb c6:L163-174 ; src/gross_pitaevskii_2D.py:112-151 ; the energy-functional lambda (:192)
                float ak, ap, ai; bool nrm;
                riesz_coefs(ph, ak, ap, ai, nrm);
                float g2 = 0.f, rho = 0.f;
                for (int o = 0; o < ph.n_out; ++o) {       // (complex psi: both components; first derivatives kept as [o][k])
                    rho = fmaf(U[o][0], U[o][0], rho);
#pragma unroll
                    for (int k = 0; k < D; ++k) { const float uk = U[o][1 + k]; ux_out[(int64_t)(o * D + k) * ld + m] = uk; g2 = fmaf(uk, uk, g2); }
                }
                rzk += (double)(ak * g2);
                rzp += (double)(ap * V * rho);
                if (!ph.complex_psi) rzi += (double)(ai * ipowf(fabsf(U[0][0]), ph.p + 1));
                else {                                     // |psi|^4 (p = 3) and the rotating-frame term: <L_z> = psi_r D psi_i - psi_i D psi_r, D = x d_y - y d_x
                    rzi += (double)(ai * rho * rho);
                    if constexpr (D >= 2) {
                        if (ph.omega_rot != 0.f) {
                            const float Dr = xv[0] * U[0][2] - xv[1] * U[0][1];
                            const float Di = xv[0] * U[1][2] - xv[1] * U[1][1];
                            rzl += (double)(U[0][0] * Di - U[1][0] * Dr);
                        }
                    }
                }
            }
        }
    }
    // all partial sums of the workgroup in ONE pass: wave-level shuffles per value, one LDS exchange, one barrier pair (one pair per value
    // before round 4: 6-8 of them were most of this kernel's time at the reference's batch sizes).  Same shuffle tree and the same wave
    // order per value as block_sum_256, so every sum is bit-identical to the one-at-a-time form.
    {
        constexpr int K = 11;
        __shared__ double redm[16 * K];
        const bool en = phys_needs_energy_sums(ph), rot = en && ph.complex_psi && ph.omega_rot != 0.f;
        double vals[K] = {num, den, bse, rzk, rzp, rzi, rzl, so[0], so[1], so[2], so[3]};
        const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool on = k < 3 || (k < 6 && en) || (k == 6 && rot) || (k >= 7 && k - 7 < ph.n_orth);       // (uniform)
            if (!on) continue;
            double v = vals[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((threadIdx.x & 63) == 0) redm[w * K + k] = v;
        }
        __syncthreads();
        if (threadIdx.x < K) {
            const int k = threadIdx.x;
            const bool on = k < 2 || (k == 2 && n_pde < N) || (k >= 3 && k < 6 && en) || (k == 6 && rot) || (k >= 7 && k - 7 < ph.n_orth);
            if (on) {
                double r = 0.0;
                for (int i2 = 0; i2 < nw; ++i2) r += redm[i2 * K + k];
                double* dst = k == 0 ? &sums[S_NUM] : k == 1 ? &sums[S_DEN] : k == 2 ? &lsums[LS_BC_SE2] : k < 7 ? &sums[S_RZ_K + (k - 3)] : &sums[S_ORTH0 + (k - 7)];
                if (k != 2 || r != 0.0) atomicAdd(dst, r);
            }
        }
    }
}

// Data-parallel steps whose forward kernel ran the head (HeadArgs): the per-workgroup (num, den, bse) triples are added here -- same
// fixed tree as the consumers below use, so the local sums are the single-GPU sums bit for bit -- and filed in sums / lsums, where the
// all-reduce of the step sums picks them up.  One wave.
__global__ __launch_bounds__(64) void k_slots_to_sums(const double* __restrict__ slots, int nslots, double* __restrict__ sums,
                                                      double* __restrict__ lsums) {
    double sv[8][3];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int b = threadIdx.x + 64 * k;
#pragma unroll
        for (int i = 0; i < 3; ++i) sv[k][i] = b < nslots ? slots[(size_t)b * 4 + i] : 0.0;
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
    if (threadIdx.x == 0) {
        sums[S_NUM] = t3[0]; sums[S_DEN] = t3[1];
        if (t3[2] != 0.0) lsums[LS_BC_SE2] += t3[2];
    }
}

// ---- phase 2: residual + seeds -------------------------------------------------------------------
// lambda = num/den (global sums), r = Hu - lambda u, sum r^2 -> gtail[GT_SUM_R2]; Ob = dLoss/dO.
template <int C, int E>
__global__ __launch_bounds__(1024) void k_seed_pde(Phys ph, Pts x, const float* __restrict__ Vpre,
                                                  const float* const* __restrict__ orth,
                                                  const float* __restrict__ u_in, const float* __restrict__ Hu_in,
                                                  const float* __restrict__ ux_in,
                                                  const double* __restrict__ sums, float* __restrict__ Ob,
                                                  float* __restrict__ resid_out, double* __restrict__ sum_r2, int64_t N,
                                                  int64_t ld, int want_seeds, const double* __restrict__ slots, int nslots,
                                                  double* __restrict__ sums_out, double* __restrict__ lsums_out) {
    constexpr int D = C - 1 - E;
    __shared__ double red[16];
    double sr2 = 0.0;
    double num, den;
    if (slots) {      // the forward kernel ran the head (HeadArgs) and left one (num, den, bse) triple per workgroup: add them in a fixed tree
        __shared__ double tot[3];
        if (threadIdx.x < 64) {
            double sv[8][3];                                // (<= 512 slots: all 24 loads in flight at once, then added in order)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int b = threadIdx.x + 64 * k;
#pragma unroll
                for (int i = 0; i < 3; ++i) sv[k][i] = b < nslots ? slots[(size_t)b * 4 + i] : 0.0;
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
            if (threadIdx.x == 0) { tot[0] = t3[0]; tot[1] = t3[1]; tot[2] = t3[2]; }
        }
        __syncthreads();
        num = tot[0]; den = tot[1];
        if (blockIdx.x == 0 && threadIdx.x == 0) {        // ... and file the totals where k_update reads them
            sums_out[S_NUM] = num; sums_out[S_DEN] = den;
            if (tot[2] != 0.0) atomicAdd(&lsums_out[LS_BC_SE2], tot[2]);
        }
    } else { num = sums[S_NUM]; den = sums[S_DEN]; }
    // energy-functional lambda (src/gross_pitaevskii_2D.py:192) and the regularisers (:197-211): lambda is NOT the Rayleigh quotient of the
    // residual's operator on these points, so d loss / d lambda = -2 w_pde / N sum r u + d L_lambda / d lambda does not vanish (SURVEY quirk
    // Q10) and is carried to u and to grad u through d lambda / d u.  sum r u = num_R - lambda den: both sums are already global.
    const float lam_all = (float)lambda_of(ph, sums, num, den);
    float lam_bar = 0.f, regf_bar = 0.f;
    if (ph.lambda_kind == GPE_LAMBDA_ENERGY) {
        const double l = (double)lam_all;
        double lb = -2.0 * (double)ph.w_pde / ph.n_global * (num - l * den);
        if (ph.w_reg_lam != 0.f) { const double q = l * l + (double)ph.reg_lam_eps; lb += -2.0 * (double)ph.w_reg_lam * l / (q * q); }
        lam_bar = (float)(lb / den);                      // (the 1/den of d lambda / d u is folded in)
    }
    if (ph.w_reg_f != 0.f) { const double q = den / ph.n_global + (double)ph.reg_f_eps; regf_bar = (float)(-2.0 * (double)ph.w_reg_f / (q * q * ph.n_global)); }
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < N; m += (int64_t)gridDim.x * blockDim.x) {
        float lam = lam_all;
        float I = (float)den * ph.dx;
        float xv[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < ph.dim; ++k) xv[k] = pts_at(x, m, ph.dim, k);
        float V = potential_at(ph, xv, Vpre, m);
        float u[2] = {0.f, 0.f}, r[2] = {0.f, 0.f}, rb[2] = {0.f, 0.f};
        float cr = (float)(2.0 * (double)ph.w_pde / ph.n_global);
        for (int o = 0; o < ph.n_out; ++o) {
            u[o] = u_in[(int64_t)o * ld + m];
            r[o] = Hu_in[(int64_t)o * ld + m] - lam * u[o];
            rb[o] = cr * r[o];
            sr2 += (double)(r[o] * r[o]);
            if (resid_out) resid_out[m * ph.n_out + o] = r[o];
        }
        if (want_seeds) {
            float ub[2] = {0.f, 0.f};
            if (!ph.complex_psi) {
                float dint = ph.abs_power ? ph.gamma * (float)ph.p * ipowf(fabsf(u[0]), ph.p - 1)
                                          : ph.gamma * (float)ph.p * ipowf(u[0], ph.p - 1);
                ub[0] = rb[0] * (V + dint - lam);
            } else {
                float ur = u[0], ui = u[1], rho = ur * ur + ui * ui, g = ph.gamma;
                ub[0] = rb[0] * (V + g * (rho + 2.f * ur * ur) - lam) + rb[1] * (2.f * g * ur * ui);
                ub[1] = rb[1] * (V + g * (rho + 2.f * ui * ui) - lam) + rb[0] * (2.f * g * ur * ui);
            }
            float cn = ph.w_norm * 4.0f * (I - 1.0f) * ph.dx;
            for (int o = 0; o < ph.n_out; ++o) ub[o] += cn * u[o];
            for (int j = 0; j < ph.n_orth; ++j) {
                float Oj = (float)(sums[S_ORTH0 + j]) * ph.dx;
                ub[0] += ph.w_orth * 2.0f * Oj * ph.dx * orth[j][m];
            }
            float sc = ph.perturb_scale;
            for (int o = 0; o < ph.n_out; ++o) {
                float Ub[C];
                Ub[0] = ub[o];
#pragma unroll
                for (int j = 0; j < D; ++j) Ub[1 + j] = 0.f;
#pragma unroll
                for (int j = 0; j < E; ++j) Ub[1 + D + j] = -ph.kin * rb[o];
                if constexpr (D >= 1) {
                    if (ph.w_riesz != 0.f) {    // d(w_riesz E)/du, /du_k with E = (ak K + ap Pv + ai Ig) / (den | 1)
                        float ak, ap, ai; bool nrm;
                        riesz_coefs(ph, ak, ap, ai, nrm);
                        const float dnm = nrm ? (float)sums[S_DEN] : 1.0f;
                        // VARIATIONAL: the interaction sum carries fI = I^(-(p-1)/2), I = dx sum u^2 (energy of the normalised state)
                        const float fI = ph.riesz_kind == GPE_RIESZ_VARIATIONAL ? powf((float)sums[S_DEN] * ph.dx, -0.5f * (float)(ph.p - 1)) : 1.0f;
                        const float cI = ph.riesz_kind == GPE_RIESZ_VARIATIONAL ? 0.5f * (float)(ph.p + 1) : 1.0f;
                        const double Lrot = ph.complex_psi ? (double)ph.omega_rot * sums[S_RZ_L] : 0.0;
                        const float Erz = nrm ? (float)((sums[S_RZ_K] + sums[S_RZ_P] + (double)(cI * fI) * sums[S_RZ_I] - Lrot) / sums[S_DEN]) : 0.0f;
                        const float uu = u[o];
                        float dint;
                        if (!ph.complex_psi) {
                            const float sg = uu < 0.f ? -1.f : 1.f;
                            dint = ai * fI * (float)(ph.p + 1) * sg * ipowf(fabsf(uu), ph.p);
                        } else dint = ai * fI * (float)(ph.p + 1) * (u[0] * u[0] + u[1] * u[1]) * uu;       // d rho^2 / d psi_o = 4 rho psi_o
                        Ub[0] += ph.w_riesz * ((2.f * ap * V * uu + dint) - 2.f * Erz * uu) / dnm;
#pragma unroll
                        for (int k = 0; k < D; ++k) Ub[1 + k] += ph.w_riesz * 2.f * ak * ux_in[(int64_t)(o * D + k) * ld + m] / dnm;
                        if constexpr (D >= 2) {
                            if (ph.complex_psi && ph.omega_rot != 0.f) {        // - Omega <L_z> / den
                                const float Om = ph.omega_rot, wq = ph.w_riesz / dnm;
                                const int q = 1 - o;                            // the other component
                                const float Dq = xv[0] * ux_in[(int64_t)(q * D + 1) * ld + m] - xv[1] * ux_in[(int64_t)(q * D + 0) * ld + m];
                                const float sgn = o == 0 ? 1.f : -1.f;          // d<L_z>/d psi_r = D psi_i ; d<L_z>/d psi_i = -D psi_r
                                Ub[0] += wq * (-Om) * sgn * Dq;
                                Ub[1] += wq * (-Om) * sgn * (xv[1] * u[q]);     // d<L_z>/d(d_x psi_r) = y psi_i ; /d(d_x psi_i) = -y psi_r
                                Ub[2] += wq * (-Om) * sgn * (-xv[0] * u[q]);    // d<L_z>/d(d_y psi_r) = -x psi_i ; /d(d_y psi_i) = x psi_r
                            }
                        }
                    }
                }
                if constexpr (D >= 1) {
                    if (lam_bar != 0.f || regf_bar != 0.f) {       // real psi, out = 1 (checked at gpe_create)
                        const float uu = u[0];
                        const float sg = uu < 0.f ? -1.f : 1.f;
                        // den * d lambda / d u = 2 V u + gamma (p+1) |u|^p sgn u - 2 lambda u ;  den * d lambda / d u_k = 2 c u_k
                        Ub[0] += lam_bar * (2.f * V * uu + ph.gamma * (float)(ph.p + 1) * sg * ipowf(fabsf(uu), ph.p) - 2.f * lam * uu) + regf_bar * uu;
#pragma unroll
                        for (int k = 0; k < D; ++k) Ub[1 + k] += lam_bar * 2.f * ph.kin * ux_in[(int64_t)k * ld + m];
                    }
                }
                if (ph.complex_psi && ph.omega_rot != 0.f && D >= 2) {
                    float Om = ph.omega_rot;
                    if (o == 1) { Ub[2] += -Om * xv[0] * rb[0]; Ub[1] += Om * xv[1] * rb[0]; }
                    else        { Ub[2] += Om * xv[0] * rb[1];  Ub[1] += -Om * xv[1] * rb[1]; }
                }
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
                for (int c = 0; c < C; ++c) Ob[((int64_t)c * ph.n_out + o) * ld + m] = sc * Ub[c];
            }
        }
    }
    double t = block_sum_256(sr2, red);
    if (threadIdx.x == 0) atomicAdd(sum_r2, t);
}

// ---- boundary batch (value only): e = base + s*NN - target; sum e^2; Ob = w_bc*2/(cnt) * e * s / world ------
__global__ __launch_bounds__(256) void k_head_seed_bc(Phys ph, float base_norm, const float* __restrict__ xb,
                                                      const float* __restrict__ target, const float* __restrict__ O,
                                                      float* __restrict__ Ob, double* __restrict__ lsums, int64_t nb,
                                                      int64_t ld) {
    __shared__ double red[4];
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double se = 0.0;
    if (m < nb) {
        float cnt = (float)(nb * ph.n_out);
        for (int o = 0; o < ph.n_out; ++o) {
            float fenv = 1.0f;
            if (ph.envelope == GPE_ENV_SIN) { float f1, f2; envelope_at(ph, xb[m * ph.dim], fenv, f1, f2); }
            float e = ph.bc_nn_scale * fenv * O[(int64_t)o * ld + m];
            if (ph.base_mode >= 0 && o == 0 && ph.base_kind != GPE_BASE_PRECOMPUTED) {
                float phi, p1, p2;
                base_at(ph, xb[m * ph.dim], base_norm, phi, p1, p2);
                e += phi;
            }
            if (target) e -= target[m * ph.n_out + o];
            se += (double)(e * e);
            Ob[(int64_t)o * ld + m] = ph.w_bc * 2.0f / cnt * e * ph.bc_nn_scale * fenv * ph.inv_world;
        }
    }
    double t = block_sum_256(se, red);
    if (threadIdx.x == 0) atomicAdd(&lsums[LS_BC_SE2], t);
}

// ---- symmetry batch: points [x ; -x] (2N, value only): diff = o(x) - sign*o(-x) ----------------------------
__global__ __launch_bounds__(256) void k_head_sym(Phys ph, const float* __restrict__ O, double* __restrict__ sums,
                                                  int64_t N, int64_t ld) {
    __shared__ double red[4];
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (m < N) {
        for (int o = 0; o < ph.n_out; ++o) {
            float d = O[(int64_t)o * ld + m] - ph.sym_sign * O[(int64_t)o * ld + N + m];
            s += (double)(d * d);
        }
    }
    double t = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(&sums[S_SYM], t);
}

__global__ __launch_bounds__(256) void k_seed_sym(Phys ph, const float* __restrict__ O, float* __restrict__ Ob,
                                                  int64_t N, int64_t ld) {
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m < N) {
        float c = (float)(2.0 * (double)ph.w_sym / ph.n_global);
        for (int o = 0; o < ph.n_out; ++o) {
            float d = O[(int64_t)o * ld + m] - ph.sym_sign * O[(int64_t)o * ld + N + m];
            Ob[(int64_t)o * ld + m] = c * d;
            Ob[(int64_t)o * ld + N + m] = -ph.sym_sign * c * d;
        }
    }
}

// ---- pre-training: loss = mean((NN - target)^2)  (refine/harmonic_pinn_simulation.py:667-668) -----------------------------
__global__ __launch_bounds__(256) void k_seed_mse(Phys ph, const float* __restrict__ x, const float* __restrict__ target,
                                                  const float* __restrict__ O, float* __restrict__ Ob,
                                                  double* __restrict__ acc, int64_t N, int64_t ld) {
    __shared__ double red[4];
    double s = 0.0;
    const int n_out = ph.n_out;
    const float c = (float)(2.0 / (ph.n_global * n_out));
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < N; m += (int64_t)gridDim.x * 256) {
        float fenv = 1.0f;
        if (ph.envelope == GPE_ENV_SIN) { float f1, f2; envelope_at(ph, x[m * ph.dim], fenv, f1, f2); }
        for (int o = 0; o < n_out; ++o) {
            float e = fenv * O[(int64_t)o * ld + m] - target[m * n_out + o];       // model.forward includes the factor
            s += (double)(e * e);
            Ob[(int64_t)o * ld + m] = c * e * fenv;
        }
    }
    double t = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, t);
}

// x -> [x ; -x]
__global__ void k_make_sym_points(const float* __restrict__ x, float* __restrict__ xs, int64_t N, int dim) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N * dim) { float v = x[i]; xs[i] = v; xs[N * dim + i] = -v; }
}

// out [n,out] row-major from O[0][o][m]
__global__ void k_copy_values(Phys ph, const float* __restrict__ x, const float* __restrict__ O, float* __restrict__ out,
                              int64_t N, int64_t ld, int n_out) {
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m < N) {
        float fenv = 1.0f;
        if (ph.envelope == GPE_ENV_SIN) { float f1, f2; envelope_at(ph, x[m * ph.dim], fenv, f1, f2); }
        for (int o = 0; o < n_out; ++o) out[m * n_out + o] = fenv * O[(int64_t)o * ld + m];
    }
}
// jets [C][n][out] from O [C][out][ld]
__global__ void k_copy_jets(const float* __restrict__ O, float* __restrict__ jets, int64_t N, int64_t ld, int n_out, int C) {
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m < N) for (int c = 0; c < C; ++c) for (int o = 0; o < n_out; ++o)
        jets[((int64_t)c * N + m) * n_out + o] = O[((int64_t)c * n_out + o) * ld + m];
}
__global__ void k_copy_psi(const float* __restrict__ u, float* __restrict__ out, int64_t N, int64_t ld, int n_out) {
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m < N) for (int o = 0; o < n_out; ++o) out[m * n_out + o] = u[(int64_t)o * ld + m];
}

// eval path: u = base + scale*NN ; sum u^2
__global__ __launch_bounds__(256) void k_eval_u(Phys ph, float base_norm, const float* __restrict__ x,
                                                const float* __restrict__ O, float* __restrict__ u, double* __restrict__ acc,
                                                int64_t N, int64_t ld) {
    __shared__ double red[4];
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (m < N) {
        for (int o = 0; o < ph.n_out; ++o) {
            float v = ph.perturb_scale * O[(int64_t)o * ld + m];
            if (ph.envelope == GPE_ENV_SIN) { float f, f1, f2; envelope_at(ph, x[m * ph.dim], f, f1, f2); v *= f; }
            if (ph.base_mode >= 0 && o == 0) {
                float phi, p1, p2;
                base_at(ph, x[m * ph.dim], base_norm, phi, p1, p2);
                v += phi;
            }
            u[(int64_t)o * ld + m] = v;
            s += (double)(v * v);
        }
    }
    double t = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, t);
}
__global__ void k_eval_finish(const float* __restrict__ u, const double* __restrict__ acc, float dx, int abs_flag,
                              float* __restrict__ u_out, float* __restrict__ dens, int64_t N, int64_t ld, int n_out) {
    int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m < N) {
        float nrm = sqrtf((float)(*acc) * dx);
        float d = 0.f;
        for (int o = 0; o < n_out; ++o) {
            float v = u[(int64_t)o * ld + m] / nrm;
            if (abs_flag) v = fabsf(v);
            if (u_out) u_out[m * n_out + o] = v;
            d += v * v;
        }
        if (dens) dens[m] = d;
    }
}
