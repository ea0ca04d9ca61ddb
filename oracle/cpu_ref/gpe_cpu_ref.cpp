// oracle/cpu_ref/gpe_cpu_ref.cpp -- CPU ORACLE / CPU BASELINE.  TEST INFRASTRUCTURE ONLY (never linked into libgpe_hip.so; only
// tests/, bench.py's cpu_baseline leg and __graft_entry__ may build or call it).
//
// Plain C++ / OpenMP restatement, in fp32, of the algorithm the HIP kernels implement for the epoch body of
// /root/reference/Gross-Pitaevskii/src/final/refine/harmonic_pinn_simulation.py:328-358 (nb c10:L85-100) on a d-dimensional
// harmonic trap (2D/3D Laplacian template: src/gross_pitaevskii_2D.py:183-195 with quirk Q1 fixed):
//   forward-mode jets (value, d first derivatives, Laplacian) through the tanh MLP        (:121-125, :158-172)
//   u, H u = -c lap u + V u + gamma u^p, Rayleigh quotient mu = sum u Hu / sum u^2         (:181-188)
//   residual MSE + w_norm (dx sum u^2 - 1)^2                                               (:191-194, :212-217)
//   hand-derived reverse pass -> d loss / d theta (mu treated as a constant: SURVEY quirk Q10)   (:358)
// Scope: real psi (one output), harmonic potential, collocation batch only (no boundary / symmetry / base terms) -- the
// like-for-like CPU cost of the work bench.py times on the GPU, and a second, independently written check of the jet algebra
// (tests/test_cpu_ref.py compares it with oracle/gpe_oracle.py).  Points are processed in blocks of PB; a block's jets live
// in thread-local buffers [feature][channel*PB] so that every inner loop is a unit-stride fused multiply-add over channel*PB.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {
constexpr int PB = 32;          // points per block
constexpr int MAXL = 12;

struct Net {
    int n_lin, dim, width[MAXL], offW[MAXL], offB[MAXL], P, maxw;
    float shift;
};

struct Work {                   // per-thread buffers
    std::vector<float> A[MAXL];     // activation jets of hidden layer l: [width][C*PB]   (A[0] is the input layer's jets)
    std::vector<float> S[MAXL];     // stored (t, z_k, z_L) of hidden layer l
    std::vector<float> Zb, Ab, Ab2;
    std::vector<double> g;
};

inline float ipow(float u, int p) { float r = 1.f; for (int i = 0; i < p; ++i) r *= u; return r; }

// The three products of a layer, register-blocked four output rows at a time so that each loaded operand element feeds four FMAs.
// Z[n][cp] = sum_k W[n][k] A[k][cp]
void gemm_fwd(const float* W, int No, int K, const float* A, float* Z, int CP) {
    int n = 0;
    for (; n + 4 <= No; n += 4) {
        float* z0 = Z + (size_t)n * CP; float* z1 = z0 + CP; float* z2 = z1 + CP; float* z3 = z2 + CP;
        for (int i = 0; i < CP; ++i) { z0[i] = 0.f; z1[i] = 0.f; z2[i] = 0.f; z3[i] = 0.f; }
        for (int k = 0; k < K; ++k) {
            const float w0 = W[n * K + k], w1 = W[(n + 1) * K + k], w2 = W[(n + 2) * K + k], w3 = W[(n + 3) * K + k];
            const float* a = A + (size_t)k * CP;
#pragma omp simd
            for (int i = 0; i < CP; ++i) { const float av = a[i]; z0[i] += w0 * av; z1[i] += w1 * av; z2[i] += w2 * av; z3[i] += w3 * av; }
        }
    }
    for (; n < No; ++n) {
        float* z = Z + (size_t)n * CP;
        for (int i = 0; i < CP; ++i) z[i] = 0.f;
        for (int k = 0; k < K; ++k) {
            const float w = W[n * K + k];
            const float* a = A + (size_t)k * CP;
#pragma omp simd
            for (int i = 0; i < CP; ++i) z[i] += w * a[i];
        }
    }
}
// Ab[k][cp] = sum_n W[n][k] Zb[n][cp]
void gemm_bwd_data(const float* W, int No, int K, const float* Zb, float* Ab, int CP) {
    int k = 0;
    for (; k + 4 <= K; k += 4) {
        float* a0 = Ab + (size_t)k * CP; float* a1 = a0 + CP; float* a2 = a1 + CP; float* a3 = a2 + CP;
        for (int i = 0; i < CP; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; a3[i] = 0.f; }
        for (int n = 0; n < No; ++n) {
            const float* wr = W + n * K + k;
            const float w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
            const float* z = Zb + (size_t)n * CP;
#pragma omp simd
            for (int i = 0; i < CP; ++i) { const float zv = z[i]; a0[i] += w0 * zv; a1[i] += w1 * zv; a2[i] += w2 * zv; a3[i] += w3 * zv; }
        }
    }
    for (; k < K; ++k) {
        float* a = Ab + (size_t)k * CP;
        for (int i = 0; i < CP; ++i) a[i] = 0.f;
        for (int n = 0; n < No; ++n) {
            const float w = W[n * K + k];
            const float* z = Zb + (size_t)n * CP;
#pragma omp simd
            for (int i = 0; i < CP; ++i) a[i] += w * z[i];
        }
    }
}
// gW[n][k] += sum_cp Zb[n][cp] A[k][cp]
void gemm_bwd_weight(double* gW, int No, int K, const float* Zb, const float* A, int CP) {
    for (int n = 0; n < No; ++n) {
        const float* z = Zb + (size_t)n * CP;
        int k = 0;
        for (; k + 4 <= K; k += 4) {
            const float* a0 = A + (size_t)k * CP; const float* a1 = a0 + CP; const float* a2 = a1 + CP; const float* a3 = a2 + CP;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma omp simd reduction(+ : s0, s1, s2, s3)
            for (int i = 0; i < CP; ++i) { const float zv = z[i]; s0 += zv * a0[i]; s1 += zv * a1[i]; s2 += zv * a2[i]; s3 += zv * a3[i]; }
            gW[n * K + k] += (double)s0; gW[n * K + k + 1] += (double)s1; gW[n * K + k + 2] += (double)s2; gW[n * K + k + 3] += (double)s3;
        }
        for (; k < K; ++k) {
            const float* a = A + (size_t)k * CP;
            float s = 0.f;
#pragma omp simd reduction(+ : s)
            for (int i = 0; i < CP; ++i) s += z[i] * a[i];
            gW[n * K + k] += (double)s;
        }
    }
}

// forward of one block: fills w.A[l], w.S[l] (if keep), returns output jets O[C][PB]
void forward_block(const Net& nt, const float* th, const float* xb /*[PB][dim]*/, int np, Work& w, float* O, bool keep) {
    const int d = nt.dim, C = d + 2, CP = C * PB, L = nt.n_lin - 1;
    // input jets: features = coordinates
    float* A0 = w.A[0].data();
    memset(A0, 0, sizeof(float) * (size_t)d * CP);
    for (int k = 0; k < d; ++k) {
        for (int p = 0; p < PB; ++p) A0[(size_t)k * CP + p] = p < np ? xb[p * d + k] : xb[(np - 1) * d + k];
        for (int p = 0; p < PB; ++p) A0[(size_t)k * CP + (1 + k) * PB + p] = 1.f;
    }
    for (int l = 0; l < L; ++l) {
        const int K = nt.width[l], No = nt.width[l + 1];
        float* Z = w.S[l + 1].data();
        gemm_fwd(th + nt.offW[l], No, K, w.A[l].data(), Z, CP);
        float* An = w.A[l + 1].data();
        const float* b = th + nt.offB[l];
        for (int n = 0; n < No; ++n) {
            float* z = Z + (size_t)n * CP;
            float* a = An + (size_t)n * CP;
            for (int p = 0; p < PB; ++p) {
                const float t = tanhf(z[p] + b[n]);
                const float s = 1.f - t * t, w2 = -2.f * t * s;
                float S2 = 0.f;
                for (int k = 0; k < d; ++k) { const float zk = z[(1 + k) * PB + p]; S2 += zk * zk; a[(1 + k) * PB + p] = s * zk; }
                a[(1 + d) * PB + p] = s * z[(1 + d) * PB + p] + w2 * S2;
                a[p] = t + nt.shift;
                if (keep) z[p] = t;          // stored: (t, z_k, z_L)
            }
        }
    }
    const float* Wo = th + nt.offW[L];
    const int K = nt.width[L];
    for (int i = 0; i < CP; ++i) O[i] = 0.f;
    for (int k = 0; k < K; ++k) {
        const float wv = Wo[k];
        const float* a = w.A[L].data() + (size_t)k * CP;
#pragma omp simd
        for (int i = 0; i < CP; ++i) O[i] += wv * a[i];
    }
    for (int p = 0; p < PB; ++p) O[p] += th[nt.offB[L]];
}
}  // namespace

extern "C" {

// scalars out: [0] loss [1] pde [2] norm [3] mu [4] num [5] den [6] sum_r2 [7] integral
// returns 0, or -1 on an unsupported problem description
int cpu_ref_loss_grad(const int* layers, int n_layers, int activation_shift, const float* theta, const float* x, int64_t N, float kin,
                      float pot_scale, const float* omega, float gamma, int p, float w_pde, float w_norm, float dx,
                      int64_t n_global, int threads, double* scalars, float* grad /* may be NULL: scalars only */) {
    if (n_layers < 3 || n_layers > MAXL || layers[n_layers - 1] != 1 || layers[0] < 1 || layers[0] > 3 || N < 1 || p < 1) return -1;
    Net nt;
    nt.n_lin = n_layers - 1; nt.dim = layers[0]; nt.shift = activation_shift ? 1.f : 0.f;
    int off = 0; nt.maxw = 0;
    for (int i = 0; i < n_layers; ++i) { nt.width[i] = layers[i]; if (layers[i] > nt.maxw) nt.maxw = layers[i]; }
    for (int j = 0; j < nt.n_lin; ++j) { nt.offW[j] = off; off += nt.width[j] * nt.width[j + 1]; nt.offB[j] = off; off += nt.width[j + 1]; }
    nt.P = off;
    const int d = nt.dim, C = d + 2, CP = C * PB, L = nt.n_lin - 1;
    const int64_t nblk = (N + PB - 1) / PB;
    const double Ng = n_global > 0 ? (double)n_global : (double)N;
#ifdef _OPENMP
    const int nthr = threads > 0 ? threads : omp_get_max_threads();
#else
    const int nthr = 1;
#endif
    std::vector<Work> works(nthr);
    for (auto& w : works) {
        for (int l = 0; l <= L; ++l) { w.A[l].assign((size_t)nt.width[l] * CP, 0.f); w.S[l].assign((size_t)nt.width[l] * CP, 0.f); }
        w.Zb.assign((size_t)nt.maxw * CP, 0.f); w.Ab.assign((size_t)nt.maxw * CP, 0.f); w.Ab2.assign((size_t)nt.maxw * CP, 0.f);
        w.g.assign(nt.P, 0.0);
    }
    // ---- pass 1: u, Hu -> num, den ------------------------------------------------------------------------------------------
    double num = 0.0, den = 0.0;
#pragma omp parallel for num_threads(nthr) reduction(+ : num, den) schedule(static)
    for (int64_t b = 0; b < nblk; ++b) {
#ifdef _OPENMP
        Work& w = works[omp_get_thread_num()];
#else
        Work& w = works[0];
#endif
        const int np = (int)((b + 1) * PB <= N ? PB : N - b * PB);
        float O[5 * PB];
        forward_block(nt, theta, x + (size_t)b * PB * d, np, w, O, false);
        for (int q = 0; q < np; ++q) {
            const float* xp = x + ((size_t)b * PB + q) * d;
            float V = 0.f;
            for (int k = 0; k < d; ++k) { const float t = omega[k] * xp[k]; V += t * t; }
            V *= pot_scale;
            const float u = O[q], lap = O[(1 + d) * PB + q];
            const float Hu = -kin * lap + V * u + gamma * ipow(u, p);
            num += (double)(u * Hu); den += (double)(u * u);
        }
    }
    const float lam = (float)(num / den);
    const float I = (float)den * dx;
    // ---- pass 2: residual, seeds, reverse pass --------------------------------------------------------------------------------
    double sr2 = 0.0;
#pragma omp parallel for num_threads(nthr) reduction(+ : sr2) schedule(static)
    for (int64_t b = 0; b < nblk; ++b) {
#ifdef _OPENMP
        Work& w = works[omp_get_thread_num()];
#else
        Work& w = works[0];
#endif
        const int np = (int)((b + 1) * PB <= N ? PB : N - b * PB);
        float O[5 * PB], Ob[5 * PB];
        forward_block(nt, theta, x + (size_t)b * PB * d, np, w, O, true);
        for (int i = 0; i < CP; ++i) Ob[i] = 0.f;
        const float cr = (float)(2.0 * (double)w_pde / Ng);
        for (int q = 0; q < np; ++q) {
            const float* xp = x + ((size_t)b * PB + q) * d;
            float V = 0.f;
            for (int k = 0; k < d; ++k) { const float t = omega[k] * xp[k]; V += t * t; }
            V *= pot_scale;
            const float u = O[q], lap = O[(1 + d) * PB + q];
            const float Hu = -kin * lap + V * u + gamma * ipow(u, p);
            const float r = Hu - lam * u;
            sr2 += (double)(r * r);
            const float rb = cr * r;
            Ob[q] = rb * (V + gamma * (float)p * ipow(u, p - 1) - lam) + w_norm * 4.f * (I - 1.f) * dx * u;
            Ob[(1 + d) * PB + q] = -kin * rb;
        }
        if (!grad) continue;
        double* g = w.g.data();
        // output map
        {
            const int K = nt.width[L];
            const float* Wo = theta + nt.offW[L];
            float* Ab = w.Ab.data();
            for (int k = 0; k < K; ++k) {
                const float* a = w.A[L].data() + (size_t)k * CP;
                float s = 0.f;
#pragma omp simd reduction(+ : s)
                for (int i = 0; i < CP; ++i) s += Ob[i] * a[i];
                g[nt.offW[L] + k] += (double)s;
                float* ab = Ab + (size_t)k * CP;
                const float wv = Wo[k];
#pragma omp simd
                for (int i = 0; i < CP; ++i) ab[i] = wv * Ob[i];
            }
            float sb = 0.f;
            for (int q = 0; q < PB; ++q) sb += Ob[q];
            g[nt.offB[L]] += (double)sb;
        }
        float* Ab = w.Ab.data();
        float* Ab2 = w.Ab2.data();
        for (int l = L - 1; l >= 0; --l) {      // hidden layer l+1 (output of linear map l)
            const int K = nt.width[l], No = nt.width[l + 1];
            float* Zb = w.Zb.data();
            const float* S = w.S[l + 1].data();
            for (int n = 0; n < No; ++n) {
                const float* st = S + (size_t)n * CP;
                const float* ab = Ab + (size_t)n * CP;
                float* zb = Zb + (size_t)n * CP;
                for (int q = 0; q < PB; ++q) {
                    const float t = st[q], s = 1.f - t * t, w2 = -2.f * t * s, qq = s * (4.f - 6.f * s);
                    const float aLb = ab[(1 + d) * PB + q], zL = st[(1 + d) * PB + q];
                    float S2 = 0.f, dot = 0.f;
                    for (int k = 0; k < d; ++k) { const float zk = st[(1 + k) * PB + q]; S2 += zk * zk; dot += zk * ab[(1 + k) * PB + q]; }
                    for (int k = 0; k < d; ++k) zb[(1 + k) * PB + q] = s * ab[(1 + k) * PB + q] + 2.f * w2 * aLb * st[(1 + k) * PB + q];
                    zb[(1 + d) * PB + q] = s * aLb;
                    zb[q] = s * ab[q] + w2 * dot + (qq * S2 + w2 * zL) * aLb;
                }
            }
            gemm_bwd_weight(g + nt.offW[l], No, K, Zb, w.A[l].data(), CP);
            for (int n = 0; n < No; ++n) {
                float sb = 0.f;
                for (int q = 0; q < PB; ++q) sb += Zb[(size_t)n * CP + q];
                g[nt.offB[l] + n] += (double)sb;
            }
            if (l > 0) { gemm_bwd_data(theta + nt.offW[l], No, K, Zb, Ab2, CP); float* t2 = Ab; Ab = Ab2; Ab2 = t2; }
        }
    }
    const double pde = sr2 / Ng, nrm = ((double)I - 1.0) * ((double)I - 1.0);
    if (scalars) {
        scalars[0] = w_pde * pde + w_norm * nrm; scalars[1] = pde; scalars[2] = nrm; scalars[3] = lam;
        scalars[4] = num; scalars[5] = den; scalars[6] = sr2; scalars[7] = I;
    }
    if (grad) {
        for (int i = 0; i < nt.P; ++i) {
            double s = 0.0;
            for (auto& w : works) s += w.g[i];
            grad[i] = (float)s;
        }
    }
    return 0;
}

int cpu_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
}
