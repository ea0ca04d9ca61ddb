// oracle/cpu_ref/selftest.cpp -- driver for the -fsanitize=address,undefined build of gpe_cpu_ref.cpp (CPU test job only).
// Runs ragged sizes in 1D/2D/3D with several thread counts and checks a handful of gradient entries against central finite
// differences of the loss (fp32: tolerance 2e-2 relative on entries that are not tiny).  TEST INFRASTRUCTURE ONLY.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

extern "C" int cpu_ref_loss_grad(const int* layers, int n_layers, int activation_shift, const float* theta, const float* x, int64_t N,
                                 float kin, float pot_scale, const float* omega, float gamma, int p, float w_pde, float w_norm,
                                 float dx, int64_t n_global, int threads, double* scalars, float* grad);

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFFFF) / 16777216.0f - 0.5f; }

int main() {
    const int cases[][6] = {{1, 16, 16, 1, 0, 0}, {2, 24, 24, 24, 1, 0}, {3, 16, 16, 1, 0, 0}};
    const int nl[] = {4, 5, 4};
    const int64_t Ns[] = {1, 31, 33, 100};
    const float omega[3] = {1.0f, 1.4f, 2.0f};
    int bad = 0;
    for (int ci = 0; ci < 3; ++ci)
        for (int64_t N : Ns)
            for (int thr : {1, 3}) {
                const int* L = cases[ci];
                const int d = L[0];
                int P = 0;
                for (int i = 0; i + 1 < nl[ci]; ++i) P += L[i] * L[i + 1] + L[i + 1];
                unsigned seed = 12345u + ci * 77 + (unsigned)N;
                std::vector<float> th(P), x((size_t)N * d), g(P), g2(P);
                for (auto& v : th) v = 0.8f * frand(seed);
                for (auto& v : x) v = 4.0f * frand(seed);
                double sc[8], s1[8], s2[8];
                if (cpu_ref_loss_grad(L, nl[ci], ci == 1, th.data(), x.data(), N, 0.5f, 0.5f, omega, 10.0f, 3, 1.0f, 20.0f, 0.05f, 0, thr, sc,
                                      g.data()) != 0) { printf("rc != 0\n"); return 2; }
                if (!isfinite(sc[0])) { printf("non-finite loss\n"); return 3; }
                for (int t = 0; t < 4; ++t) {       // central differences (the Rayleigh quotient moves with theta; its branch is zero: Q10)
                    const int i = (int)(((unsigned)(frand(seed) * 1e6f + 5e5f)) % (unsigned)P);
                    const float keep = th[i], h = 2e-3f;
                    th[i] = keep + h; cpu_ref_loss_grad(L, nl[ci], ci == 1, th.data(), x.data(), N, 0.5f, 0.5f, omega, 10.0f, 3, 1.0f, 20.0f, 0.05f, 0, thr, s1, nullptr);
                    th[i] = keep - h; cpu_ref_loss_grad(L, nl[ci], ci == 1, th.data(), x.data(), N, 0.5f, 0.5f, omega, 10.0f, 3, 1.0f, 20.0f, 0.05f, 0, thr, s2, nullptr);
                    th[i] = keep;
                    const double fd = (s1[0] - s2[0]) / (2.0 * h);
                    const double tol = 3e-2 * fmax(fabs(fd), fabs((double)g[i])) + 2e-3 * fabs(sc[0]);
                    if (fabs(fd - g[i]) > tol) { printf("case %d N %ld thr %d entry %d: grad %g fd %g\n", ci, (long)N, thr, i, g[i], fd); ++bad; }
                }
            }
    if (cpu_ref_loss_grad(cases[0], 2, 0, nullptr, nullptr, 4, 1, 1, omega, 0, 3, 1, 1, 1, 0, 1, nullptr, nullptr) != -1) { printf("bad args accepted\n"); return 4; }
    printf("selftest: %d finite-difference mismatches\n", bad);
    return bad ? 1 : 0;
}
