// gpe_engine.hip -- step orchestrator + C ABI (include/gpe_hip.h) of libgpe_hip.so.  gfx950 only.
//
// One engine = one GPU = one stream.  A training step (the epoch body of
// refine/harmonic_pinn_simulation.py:328-361 / nb c10:L84-103) is three phases:
//   begin    : [pack weights] zero accumulators; jet forward on the collocation batch; head (u, Hu, sums)
//              [+ symmetry batch forward]                        -> "sums" exchange buffer (double[8])
//   backward : lambda, residual, seeds; reverse pass; boundary batch fwd+bwd [+ symmetry bwd]
//                                                                 -> "grad" exchange buffer (float[P+4])
//   update   : grad-norm clip, Adam, LR scheduler, history record (one single-workgroup kernel)
// With world_size > 1 the caller all-reduces the two exchange buffers between the phases (RCCL).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

// RCCL is dlopen'ed at gpe_comm_init: no link-time dependency, and no build-time one either -- the handful of types and enum
// values the five entry points need are declared here as rccl.h (NCCL 2.x ABI) declares them.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclFloat = 7, ncclDouble = 8 } ncclDataType_t;
}

#include "gpe_common.h"
#include "gpe_head.h"
#include "gpe_generic.h"
#include "gpe_fused.h"
#include "gpe_wide_api.h"

static thread_local std::string g_create_error;

#define HIPCHK(e, call)                                                                          \
    do {                                                                                         \
        hipError_t _st = (call);                                                                 \
        if (_st != hipSuccess) {                                                                 \
            char _b[512];                                                                        \
            snprintf(_b, sizeof _b, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_st)); \
            (e)->err = _b;                                                                       \
            return GPE_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define FAIL(e, code, ...)                                \
    do {                                                  \
        char _b[512];                                     \
        snprintf(_b, sizeof _b, __VA_ARGS__);             \
        (e)->err = _b;                                    \
        return (code);                                    \
    } while (0)

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------------
// update kernel: clip_grad_norm_ + Adam + scheduler + record   (refine/...:359-361, 364-381)
// ------------------------------------------------------------------------------------------------
// Large parameter vectors (P >= UPD_MULTI_MIN: [2,128x5,1] and up) run the update on UPD_G workgroups -- one workgroup took 0.40 ms
// per step at P = 330 241.  k_update_part: per-workgroup partial sums of |g|^2 over contiguous chunks (fixed order: deterministic
// for a given P) + a snapshot of the step sums and of the optimiser state; k_update<true>: every workgroup adds the partial sums in the
// same order and derives the same clip factor / step size from the SNAPSHOTS (workgroup 0 alone writes the optimiser state, the
// record and the history, and zeroes the step sums), then updates its chunk.  The repacking of the hidden-hidden weights needs ALL
// updated parameters, so it is left to the next step's k_begin in this mode.
#define UPD_G 64
#define UPD_MULTI_MIN 32768
struct UpdSnap { double sums[S_COUNT]; double lsums[LS_COUNT]; OptDev od; double part[UPD_G]; };
GPE_DEV int upd_chunk(int P) { return (((P + UPD_G - 1) / UPD_G) + 3) & ~3; }
static int upd_chunk_host(int P) { return (((P + UPD_G - 1) / UPD_G) + 3) & ~3; }
__global__ __launch_bounds__(1024) void k_update_part(int P, const float* __restrict__ grad, const double* __restrict__ sums,
                                                       const double* __restrict__ lsums, const OptDev* __restrict__ od,
                                                       UpdSnap* __restrict__ snap) {
    __shared__ double red[16];
    const int chunk = upd_chunk(P), lo = blockIdx.x * chunk, hi = min(P, lo + chunk);
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 1024) { double g = grad[i]; acc += g * g; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < 16; ++i) tot += red[i];
        snap->part[blockIdx.x] = tot;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < S_COUNT) snap->sums[threadIdx.x] = sums[threadIdx.x];
        if (threadIdx.x < LS_COUNT) snap->lsums[threadIdx.x] = lsums[threadIdx.x];
        if (threadIdx.x == 64) snap->od = *od;
    }
}

// one Adam element, torch's op order (refine/...:359-361): shared by every update loop so that they agree bit for bit
GPE_DEV void adam_element(float graw, float& m, float& v, float& th, float coef, float ss, float b2s, float b1, float b2, float eps) {
#pragma clang fp contract(off)      // torch's kernels round every product before the add; and every call site must round alike
    const float g = graw * coef;
    m = m + (g - m) * (1.0f - b1);                 // exp_avg.lerp_(grad, 1-beta1)
    v = v * b2 + (1.0f - b2) * g * g;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float denom = sqrtf(v) / b2s + eps;
    th = th - ss * (m / denom);                    // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// gradient element as the update reads it.  SC1 (the update runs inside the slab-reduction launch, k_reduce_update below): the values
// were stored by OTHER workgroups of this launch with write-through (sc1) stores -- read them past this CU's L1
template <bool SC1>
GPE_DEV float grad_load(const float* p) {
    if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

// W_j[n][kk] of a hidden-hidden map -> its slots in the two packed copies (pack_weight_element inverted): the update SCATTERS the new
// weights instead of a gather pass behind a barrier
GPE_DEV void scatter_pack(const NetDesc& nd, int H, int i, float th, float* __restrict__ Wpk, float* __restrict__ WpkT) {
    const int lg = H == 64 ? 6 : (H == 32 ? 5 : (H == 128 ? 7 : 8)), NT = H >> 4, maps = nd.n_lin - 2;
    for (int j = 1; j <= maps; ++j) {
        const int e = i - nd.offW[j];
        if (e >= 0 && e < H * H) {
            const int n = e >> lg, kk = e & (H - 1);
            const int nt = n >> 4, kt = kk >> 4;
            Wpk[(((j - 1) * NT + nt) * NT + kt) * 256 + ((n & 15) + 16 * ((kk & 15) >> 2)) * 4 + (kk & 3)] = th;
            WpkT[(((j - 1) * NT + kt) * NT + nt) * 256 + ((kk & 15) + 16 * ((n & 15) >> 2)) * 4 + (n & 3)] = th;
        }
    }
}

template <bool MULTI, bool SC1 = false>
GPE_DEV void update_core(int P, float* __restrict__ theta, float* __restrict__ am,
                         float* __restrict__ av, const float* __restrict__ grad,
                         const double* __restrict__ sums_in, const double* __restrict__ lsums_in,
                         const Phys& ph, const OptCfg& oc, OptDev* __restrict__ od,
                         gpe_scalars* __restrict__ hist, int cap, gpe_scalars* __restrict__ last,
                         double bc_cnt, int do_update, int mse_mode, const NetDesc& nd, int H,
                         float* __restrict__ Wpk, float* __restrict__ WpkT, int n_pack,
                         double* __restrict__ dbl, int n_dbl, double* __restrict__ dbl_keep,
                         const UpdSnap* __restrict__ snap, int pack_mode) {
    // pack_mode (one workgroup): 1 = repack the hidden-hidden weights by a gather pass over the updated parameters (also writes the
    // bf16 pieces the opt-in split-bf16 kernels read); 2 = every thread SCATTERS its updated weights into the two packed copies inside
    // the Adam loop (no second pass, no barrier, no bf16 pieces).  REG_E: up to this many elements per thread are loaded ONCE, at the
    // top, into registers (gradient for the norm, Adam moments and parameters behind it): at the reference's batch sizes this kernel
    // is a chain of dependent round trips to L2 -- it took 16.6 us of a 52 us step at 4 000 points -- and every load issued early is
    // one round trip less.
    constexpr int REG_E = 13;                                 // (13 312 parameters: the 4 x 64 networks of BASELINE; 16 would spill at 128 registers)
    const bool cached = !MULTI && P <= REG_E * 1024 && !(pack_mode & 4);      // (bit 2 of pack_mode: tuning switch GPE_UPDATE_CACHE=0)
    pack_mode &= 3;
    float cg[REG_E], cm[REG_E], cv[REG_E], ct[REG_E];
    __shared__ double red[16];
    __shared__ float s_coef, s_ss, s_b2s;
    __shared__ int s_skip, s_book, s_frozen;
    __shared__ long long s_step;
    __shared__ gpe_scalars s_rec;
    const double* sums = MULTI ? snap->sums : sums_in;
    const double* lsums = MULTI ? snap->lsums : lsums_in;
    const OptDev* odr = MULTI ? &snap->od : od;                    // state the step size is derived from
    const bool lead = !MULTI || blockIdx.x == 0;                   // the workgroup that writes the optimiser state / record / history
    const int lo = MULTI ? blockIdx.x * upd_chunk(P) : 0, hi = MULTI ? min(P, lo + upd_chunk(P)) : P;
    // the scalars thread 0 needs behind the norm are requested NOW, beside the elements: one round trip less in its serial section
    double p_num = 0.0, p_den = 0.0, p_bcse = 0.0, p_lr = 0.0, p_b1p = 0.0, p_b2p = 0.0;
    float p_sr2 = 0.f;
    long long p_step = 0;
    int p_stopped = 0;
    if (threadIdx.x == 0) {
        p_num = sums[S_NUM]; p_den = sums[S_DEN]; p_bcse = lsums[LS_BC_SE2]; p_sr2 = grad_load<SC1>(&grad[P + GT_SUM_R2]);
        p_lr = odr->lr; p_b1p = odr->b1p; p_b2p = odr->b2p; p_step = odr->step; p_stopped = odr->stopped;
    }
    if constexpr (!MULTI) {
        double acc = 0.0;
        if (cached) {
#pragma unroll
            for (int k = 0; k < REG_E; ++k) { const int i = threadIdx.x + k * 1024; cg[k] = i < P ? grad_load<SC1>(&grad[i]) : 0.f; }
#pragma unroll
            for (int k = 0; k < REG_E; ++k) {
                const int i = threadIdx.x + k * 1024;
                if (i < P) { cm[k] = am[i]; cv[k] = av[i]; ct[k] = theta[i]; }
            }
#pragma unroll
            for (int k = 0; k < REG_E; ++k) { const double g = cg[k]; acc += g * g; }          // (same order as the loop below)
        } else
        for (int i = threadIdx.x; i < P; i += 1024) { double g = grad_load<SC1>(&grad[i]); acc += g * g; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double tot = 0.0;
        if constexpr (MULTI) { for (int i = 0; i < UPD_G; ++i) tot += snap->part[i]; }
        else { for (int i = 0; i < 16; ++i) tot += red[i]; }
        double gn = sqrt(tot);
        double num = p_num, den = p_den;
        double lam = (double)(float)lambda_of(ph, sums, num, den);
        double I = (double)((float)den * ph.dx);
        double sr2 = (double)p_sr2;
        double pde = sr2 / ph.n_global;
        double nrm = (I - 1.0) * (I - 1.0);
        double bc = bc_cnt > 0 ? p_bcse / bc_cnt : 0.0;
        double sym = ph.w_sym != 0.f ? sums[S_SYM] / ph.n_global : 0.0;
        double orth = 0.0;
        for (int j = 0; j < ph.n_orth; ++j) { double oj = sums[S_ORTH0 + j] * ph.dx; orth += oj * oj; }
        double riesz = 0.0;
        if (ph.w_riesz != 0.f) {          // (the sums are also filed for the energy-functional lambda; the TERM only with its weight)
            const double fI = ph.riesz_kind == GPE_RIESZ_VARIATIONAL ? pow(I, -0.5 * (double)(ph.p - 1)) : 1.0;
            const double Lrot = ph.complex_psi ? (double)ph.omega_rot * sums[S_RZ_L] : 0.0;      // rotating frame: - Omega <L_z>
            riesz = (sums[S_RZ_K] + sums[S_RZ_P] + fI * sums[S_RZ_I] - Lrot) / (ph.riesz_kind == GPE_RIESZ_SUM ? 1.0 : den);
        }
        double reg = reg_terms(ph, den, lam);
        double loss = ph.w_pde * pde + ph.w_bc * bc + ph.w_norm * nrm + ph.w_sym * sym + ph.w_orth * orth + ph.w_riesz * riesz + reg;
        if (mse_mode) {           // pre-training: loss = mean((NN - target)^2); plain Adam (no clip, no scheduler, no early stop)
            loss = (double)grad_load<SC1>(&grad[P + GT_MSE_SE2]) / (ph.n_global * ph.n_out);
            lam = 0.0; pde = 0.0; nrm = 0.0; bc = 0.0; sym = 0.0; orth = 0.0; riesz = 0.0; reg = 0.0;
        }
        int skip = !(isfinite(loss) && isfinite(gn));
        const int frozen = do_update && p_stopped;
        long long step = p_step + (do_update && !skip && !frozen ? 1 : 0);
        double lr = p_lr;
        float coef = 1.0f;
        if (oc.clip_norm > 0.f && !mse_mode) coef = (float)fmin(1.0, (double)oc.clip_norm / (gn + 1e-6));
        const bool commit = do_update && !skip && !frozen;
        const double b1p = commit ? p_b1p * (double)oc.beta1 : p_b1p, b2p = commit ? p_b2p * (double)oc.beta2 : p_b2p;
        if (commit && lead) { od->b1p = b1p; od->b2p = b2p; }
        double bc1 = 1.0 - b1p;
        double bc2 = 1.0 - b2p;
        s_coef = coef;
        s_ss = (float)(lr / bc1);
        s_b2s = (float)sqrt(bc2);
        s_skip = skip || !do_update || frozen;
        gpe_scalars r;
        r.loss = loss; r.pde = pde; r.bc = bc; r.norm = nrm; r.sym = sym; r.orth = orth; r.mu = lam;
        r.num = num; r.den = den; r.sum_r2 = sr2; r.integral = I; r.grad_norm = gn; r.lr = lr;
        r.step = (double)step; r.nonfinite = skip ? 1.0 : 0.0; r.riesz = riesz; r.reg = reg;
        s_rec = r; s_step = step; s_frozen = frozen; s_book = 1;
    }
    __syncthreads();
    // bookkeeping (record, history, early stop, scheduler: double-precision log / pow / cos) on the last thread, concurrently
    // with the Adam loop of the others
    if (threadIdx.x == 1023 && s_book && lead) {
        const gpe_scalars r = s_rec;
        const long long step = s_step;
        const int frozen = s_frozen, skip = r.nonfinite != 0.0;
        const double loss = r.loss;
        if (!frozen) *last = r;
        if (do_update && !frozen) {
            if (!skip) {
                hist[(step - 1) % cap] = r;
                od->step = step;
                if (mse_mode) { /* no scheduler / early-stop bookkeeping while pre-training */ }
                else {
                // early stopping bookkeeping (refine/...:366-372, 389-400)
                if (loss < od->es_best) { od->es_best = loss; od->es_count = 0; } else od->es_count += 1;
                if ((oc.stop_tol > 0.f && loss <= (double)oc.stop_tol) ||
                    (oc.stop_patience > 0 && od->es_count >= oc.stop_patience)) { od->stopped = 1; od->stop_step = step; }
                // scheduler.step(total_loss)
                if (oc.sched == GPE_SCHED_COSINE_LOSS) {           // quirk Q4: the loss value is the epoch
                    double epoch = (double)(float)loss, T_cur, T_i;
                    if (epoch >= oc.T_0) {
                        if (oc.T_mult == 1.0f) { T_cur = fmod(epoch, (double)oc.T_0); T_i = oc.T_0; }
                        else {
                            int n = (int)(log(epoch / oc.T_0 * (oc.T_mult - 1.0) + 1.0) / log((double)oc.T_mult));
                            T_cur = epoch - oc.T_0 * (pow((double)oc.T_mult, n) - 1.0) / (oc.T_mult - 1.0);
                            T_i = oc.T_0 * pow((double)oc.T_mult, n);
                        }
                    } else { T_i = oc.T_0; T_cur = epoch; }
                    od->lr = oc.eta_min + (od->lr0 - oc.eta_min) * (1.0 + cos(M_PI * T_cur / T_i)) / 2.0;
                } else if (oc.sched == GPE_SCHED_PLATEAU) {        // ReduceLROnPlateau(mode='min', rel threshold)
                    if (loss < od->best * (1.0 - oc.threshold)) { od->best = loss; od->num_bad = 0; }
                    else od->num_bad += 1;
                    if (od->num_bad > oc.patience) {
                        double nl = fmax(od->lr * oc.factor, (double)oc.min_lr);
                        if (od->lr - nl > 1e-8) od->lr = nl;
                        od->num_bad = 0;
                    }
                }
                }
            } else {
                od->nonfinite = 1;
            }
        }
    }
    if (!s_skip && cached) {
        const float coef = s_coef, ss = s_ss, b2s = s_b2s;
        const float b1 = oc.beta1, b2 = oc.beta2, eps = oc.eps;
#pragma unroll
        for (int k = 0; k < REG_E; ++k) {
            const int i = threadIdx.x + k * 1024;
            if (i < P) {
                float m = cm[k], v = cv[k], th = ct[k];
                adam_element(cg[k], m, v, th, coef, ss, b2s, b1, b2, eps);
                theta[i] = th;
                am[i] = m; av[i] = v;
                if (pack_mode == 2 && n_pack > 0) scatter_pack(nd, H, i, th, Wpk, WpkT);
            }
        }
    } else if (!s_skip) {
        const float coef = s_coef, ss = s_ss, b2s = s_b2s;
        const float b1 = oc.beta1, b2 = oc.beta2, eps = oc.eps;
        for (int i = lo + threadIdx.x; i < hi; i += 1024) {
            float m = am[i], v = av[i], th = theta[i];
            adam_element(grad_load<SC1>(&grad[i]), m, v, th, coef, ss, b2s, b1, b2, eps);
            theta[i] = th;
            am[i] = m; av[i] = v;
            if (MULTI && pack_mode == 2 && n_pack > 0) scatter_pack(nd, H, i, th, Wpk, WpkT);      // (small networks on the split update)
        }
    }
    // Prepare the NEXT step, so that it can start with its forward kernel: the step sums are consumed (thread 0 read them
    // before the barrier above) -> zero them; repack the hidden-hidden weights in MFMA fragment order from the new parameters.
    __syncthreads();
    // (stale-gradient mode: this step's sums are kept for the NEXT update, which applies this step's gradient)
    if (lead) for (int i = threadIdx.x; i < n_dbl; i += 1024) { if (dbl_keep) dbl_keep[i] = dbl[i]; dbl[i] = 0.0; }
    if constexpr (!MULTI) {
        if (pack_mode != 2 || !cached || s_skip)                  // (a skipped update leaves the parameters, and with them the packed copies, as they were -- but a repack is harmless)
            for (int i = threadIdx.x; i < n_pack; i += 1024) pack_weight_element(nd, H, theta, Wpk, WpkT, i);
    }
}

template <bool MULTI>
__global__ __launch_bounds__(1024) void k_update(int P, float* __restrict__ theta, float* __restrict__ am,
                                                  float* __restrict__ av, const float* __restrict__ grad,
                                                  const double* __restrict__ sums_in, const double* __restrict__ lsums_in,
                                                  Phys ph, OptCfg oc, OptDev* __restrict__ od,
                                                  gpe_scalars* __restrict__ hist, int cap, gpe_scalars* __restrict__ last,
                                                  double bc_cnt, int do_update, int mse_mode, NetDesc nd, int H,
                                                  float* __restrict__ Wpk, float* __restrict__ WpkT, int n_pack,
                                                  double* __restrict__ dbl, int n_dbl, double* __restrict__ dbl_keep,
                                                  const UpdSnap* __restrict__ snap, int pack_mode) {
    update_core<MULTI>(P, theta, am, av, grad, sums_in, lsums_in, ph, oc, od, hist, cap, last, bc_cnt, do_update, mse_mode, nd, H,
                       Wpk, WpkT, n_pack, dbl, n_dbl, dbl_keep, snap, pack_mode);
}

// closes the reverse phase: adds the boundary-batch gradient (computed on the side stream) and fills the exchange tail
__global__ void k_tail(float* __restrict__ grad, const float* __restrict__ add, int P, const double* dsc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (add && i < P) grad[i] += add[i];
    if (i == 0) { grad[P + GT_SUM_R2] = (float)dsc[0]; grad[P + GT_MSE_SE2] = (float)dsc[2]; }
}

// Slab reduction AND update in one launch (small parameter vectors, whole steps: gpe_step / gpe_run).  At the reference's batch sizes
// the update was a launch of its own whose ONE workgroup started only when the whole reduction grid had drained: 2.6 us of launch plus a
// chain of dependent L2 round trips (10.4 of a 40.5 us step at 4 000 points).  Here every workgroup reduces its 64 columns as
// k_grad_reduce does and stores them write-through (sc1: no L2 write-back fence -- the release fence in every workgroup is what made the
// round-3 attempt at this fusion slower), takes a ticket, and the workgroup whose ticket is the last runs the single-workgroup update on
// the spot, reading the gradient past its L1 (hand-off form of MI355X_MICROARCH.md, "Valid forms": sc1 stores, every storing wave's
// vmcnt(0), workgroup barrier, one agent-scope atomic add per workgroup, the last adder loads sc1 behind a barrier).  Same arithmetic in
// the same order as k_grad_reduce + k_update<false>: bit-identical results (test_update_kernel_forms_are_bit_identical).
__global__ __launch_bounds__(1024) void k_reduce_update(const float* __restrict__ gslab, int nslab, int Ppad, int P, float* __restrict__ grad,
                                                         const float* __restrict__ add, const double* __restrict__ tail_dsc,
                                                         unsigned* __restrict__ ticket,
                                                         float* __restrict__ theta, float* __restrict__ am, float* __restrict__ av,
                                                         const double* __restrict__ sums_in, const double* __restrict__ lsums_in,
                                                         Phys ph, OptCfg oc, OptDev* __restrict__ od, gpe_scalars* __restrict__ hist, int cap,
                                                         gpe_scalars* __restrict__ last, double bc_cnt, NetDesc nd, int H,
                                                         float* __restrict__ Wpk, float* __restrict__ WpkT, int n_pack,
                                                         double* __restrict__ dbl, int n_dbl, int pack_mode) {
    __shared__ float red[16][64];
    __shared__ int s_last;
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
        __hip_atomic_store(&grad[i], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(&grad[P + GT_SUM_R2], (float)tail_dsc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&grad[P + GT_MSE_SE2], (float)tail_dsc[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave: its stores have left the CU
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == gridDim.x - 1;
        if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // every other workgroup has drawn: ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    update_core<false, true>(P, theta, am, av, grad, sums_in, lsums_in, ph, oc, od, hist, cap, last, bc_cnt, 1, 0, nd, H, Wpk, WpkT, n_pack, dbl,
                             n_dbl, nullptr, nullptr, pack_mode);
}

// Slab reduction of SMALL parameter vectors for whole steps (gpe_step / gpe_run), in the partition of the multi-workgroup update: UPD_G
// workgroups, workgroup b owns the parameters [b chunk, (b+1) chunk), chunk = upd_chunk(P) <= 1024.  Besides the gradient it leaves what
// k_update_part would: the partial sums of |g|^2 (fixed order) and the snapshot of the step sums / optimiser state -- so the update that
// follows is k_update<true> on UPD_G workgroups, a handful of elements per thread, instead of ONE workgroup walking all P parameters through
// a chain of dependent L2 round trips (10.4 us of a 40.5 us step at 4 000 points).  Threads: column = t % chunk, slab group = t / chunk.
__global__ __launch_bounds__(1024) void k_grad_reduce_part(const float* __restrict__ gslab, int nslab, int Ppad, int P, float* __restrict__ grad,
                                                            const float* __restrict__ add, const double* __restrict__ tail_dsc,
                                                            const double* __restrict__ sums, const double* __restrict__ lsums,
                                                            const OptDev* __restrict__ od, UpdSnap* __restrict__ snap) {
    __shared__ float red[1024];
    __shared__ double red2[16];
    const int chunk = upd_chunk(P), lo = blockIdx.x * chunk;
    const int ng = 1024 / chunk;                                  // slab groups (>= 1: chunk <= 1024 is checked by the launcher)
    const int col = threadIdx.x % chunk, grp = threadIdx.x / chunk;
    const int i = lo + col;
    float s = 0.f;
    if (grp < ng && i < P) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;            // four loads in flight per thread (fixed order of the final sum)
        int b = grp;
        for (; b + 3 * ng < nslab; b += 4 * ng) {
            s0 += gslab[(size_t)b * Ppad + i];
            s1 += gslab[(size_t)(b + ng) * Ppad + i];
            s2 += gslab[(size_t)(b + 2 * ng) * Ppad + i];
            s3 += gslab[(size_t)(b + 3 * ng) * Ppad + i];
        }
        for (; b < nslab; b += ng) s0 += gslab[(size_t)b * Ppad + i];
        s = (s0 + s1) + (s2 + s3);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    double gsq = 0.0;
    if (grp == 0 && i < P) {
        float t = 0.f;
        for (int k = 0; k < ng; ++k) t += red[k * chunk + col];
        if (add) t += add[i];
        grad[i] = t;
        gsq = (double)t * (double)t;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gsq += __shfl_down(gsq, o, 64);
    if ((threadIdx.x & 63) == 0) red2[threadIdx.x >> 6] = gsq;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < 16; ++k) tot += red2[k];
        snap->part[blockIdx.x] = tot;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < S_COUNT) snap->sums[threadIdx.x] = sums[threadIdx.x];
        if (threadIdx.x < LS_COUNT) snap->lsums[threadIdx.x] = lsums[threadIdx.x];
        if (threadIdx.x == 64) snap->od = *od;
        if (threadIdx.x == 65) { grad[P + GT_SUM_R2] = (float)tail_dsc[0]; grad[P + GT_MSE_SE2] = (float)tail_dsc[2]; }
    }
}

// ------------------------------------------------------------------------------------------------
struct Batch {
    Pts pts = {nullptr, nullptr, 0};   // point coordinates [n][dim]: rows [0,na) from a, the rest from b (merged boundary points)
    const float* V = nullptr;
    int64_t n = 0, ld = 0;
    int C = 1, E = 0;              // channels: 1 value + D first derivatives + E second-order channels (gpe_common.h)
    float* O = nullptr;            // [C][n_out][ld]
    float* Ob = nullptr;
    float* u = nullptr;            // [n_out][ld]  (main batch)
    float* Hu = nullptr;
    float* ux = nullptr;           // [dim][ld] grad u (Riesz term)
    float* stored = nullptr;       // fused: fragment-native stored activations
    float* Z0 = nullptr;           // wide set: adjoint-jet ping-pong buffers [tile][C][H/16][256]
    float* Z1 = nullptr;
    std::vector<float*> S;         // generic: per hidden layer [C][H][ld]
    float* A0 = nullptr;           // generic adjoint ping/pong [C][maxW][ld]
    float* A1 = nullptr;
    float* A2 = nullptr;           // third adjoint buffer (residual networks: a block's output adjoint waits for the skip join)
    float* xown = nullptr;         // owned copy of points (symmetry batch)
    std::vector<void*> allocs;
};

struct gpe_engine {
    gpe_config cfg;
    NetDesc nd;
    Phys ph;
    OptCfg oc;
    int device = 0;
    hipStream_t stream = nullptr;
    int path = GPE_PATH_GENERIC;
    int H = 0;                     // uniform hidden width (fused)
    bool wide = false;             // fused path: reverse pass by the wide kernel set (gpe_wide.h): H = 256 or 128
    bool wide_fwd = false;         // ... and the forward pass too (H = 256, H = 128 in 3D; H = 128 in 1D/2D keeps f_forward_coop: measured)
    int64_t gen_min_chunk = 32;    // generic set, split-K weight gradient on the matrix cores: fewest points per wave (GPE_GEN_MIN_CHUNK; 256 before round 4)
    int64_t wide_min_tiles = 2048; // H = 128 in 1D/2D: batches below this many 16-point tiles take the single-launch cooperative reverse kernel
    int P = 0, Ppad = 0;
    // width padding: hidden widths without an MFMA kernel instance run zero-padded to the next width that has one.  P counts the
    // padded network (what the kernels, the optimiser and the gradient exchange see); the caller's flat vector has P_user entries
    // and umap[i] is the padded index of its i-th entry.  Empty map: no padding, P_user == P.
    int P_user = 0;
    std::vector<int> umap;
    std::vector<int> pad_bias;     // ShiftedTanh: biases of the padded units (held at -40: tanh = -1 exactly, so the unit outputs 0 with zero slope)
    int user_layers[GPE_MAX_LAYERS] = {0};
    float base_norm = 1.f;
    float *theta = nullptr, *am = nullptr, *av = nullptr, *grad = nullptr;   // grad: P + GT_COUNT
    double* dbl = nullptr;         // [S_COUNT sums | LS_COUNT local | 4 misc]
    OptDev* od = nullptr;
    bool update_cache = true;      // single-workgroup update: register-cached elements + scatter packing (GPE_UPDATE_CACHE=0: the two-pass form)
    bool fuse_seed = true;         // small batches: the pipelined reverse kernel forms the seeds itself (GPE_FUSE_SEED=0: k_seed_pde)
    bool seedf_now = false;        // ... for the reverse pass being enqueued
    int64_t fuse_seed_max = 65536; // ... up to this many points (beyond, the redundant seed arithmetic of the four waves costs more than the launch)
    int share_min_tiles = 16;      // ... for batches of at least this many tiles per workgroup (per wave in the forward kernel)
    int fwd_share = 640;           // f_forward: the same for its waves (GPE_FWD_SHARE; measured flat between 608 and 672: NS step 2.703 -> 2.688 ms)
    int pipe_share = 576;          // f_backward_pipe: share (/1024) of a CU's tiles for its first-dispatched workgroup; 0 = even (GPE_PIPE_SHARE;
                                   // NS reverse kernel 1.741 ms even, 1.727 / 1.719 / 1.715 / 1.730 / 1.741 ms at 544 / 576 / 592 / 608 / 640)
    bool fuse_head = true;         // ... and the cooperative forward kernel runs the head (GPE_FUSE_HEAD=0: k_head_pde): whole steps only,
    int64_t fuse_head_max = 6144;  // up to this many points (measured: 43.2 vs 46.5 us at 4 000, equal at 6 000, 64.3 vs 62.9 us at 8 192)
    bool fh_want = false;          // gpe_step / graph capture in progress: nobody reads the step sums between the passes
    bool fh_now = false;           // the forward pass of this step left per-workgroup head sums in head_slots
    int fh_nslots = 0;             // ... this many triples (= workgroups of that forward launch)
    int64_t fuse_head_tile_min = 32769;   // f_forward (per-wave tiles) runs the head from this many points on (GPE_FUSE_HEAD_TILE_MIN)
    double* head_slots = nullptr;  // [HEAD_SLOTS][4]
    bool fuse_update = false;      // small P, whole steps: the last-arriving workgroup of the slab reduction runs the update (k_reduce_update; opt-in GPE_FUSE_UPDATE=1)
    bool fu_want = false;          // gpe_step / graph capture in progress: reduction and update are enqueued back to back
    bool fu_done = false;          // this step's update already ran inside the slab reduction
    unsigned* upd_ticket = nullptr;
    bool split_update = false;     // small P, whole steps: slab reduction in the update's partition + k_update<true> on UPD_G workgroups (opt-in GPE_SPLIT_UPDATE=1)
    UpdSnap* upd_snap_small = nullptr;
    bool fu_parts = false;         // this step's slab reduction left partial norms + snapshot for the multi-workgroup update
    UpdSnap* upd_snap = nullptr;   // multi-workgroup update (P >= UPD_MULTI_MIN): partial norms + snapshot of sums / optimiser state
    gpe_scalars *hist = nullptr, *last = nullptr;
    int cap = 65536;
    float *Wpk = nullptr, *WpkT = nullptr, *gslab = nullptr;
    int nslab = 0;
    bool packed_dirty = true;
    bool acc_clean = false;                       // step sums are zero (left so by k_update): the next step needs no k_begin
    bool ext_exchange = false;
    bool fwd_wlds = false;                        // forward kernel stages the hidden-hidden weights in LDS
    bool fwd_b6 = false;                          // f_forward_b6: H x H maps as six bf16 MFMA products per fp32 product (H <= 64; GPE_FWD_B6)
    bool bwd_b6 = false;                          // f_backward_coop<..., B6>: the adjoint products the same way (H <= 64; GPE_BWD_B6)
    int coop_wg_per_cu = 2;                       // cooperative reverse kernels at H <= 64: persistent workgroups per CU (tuning: GPE_COOP_WG_PER_CU)
    int fwd_wg_per_cu = 2;                        // per-wave-tile forward kernel: persistent workgroups per CU (tuning: GPE_FWD_WG_PER_CU; 3 needs a -DGPE_FWD_WAVES=3 build)
    bool bwd_pipe = true;                         // cooperative reverse kernel in its one-barrier-per-map form (f_backward_pipe; H <= 64; GPE_PIPE)
    bool bwd_racc = false;                        // reverse kernel keeps the H x H weight gradients in registers (1 wave/SIMD)
    int nslab_g = 16;                             // H = 128: number of global-atomic gradient slabs
    int coop = 1;                                 // cooperative reverse kernel: 0 never, 1 whenever compiled for the shape, -1 by batch size
    int64_t coop_max_tiles = 0;
    bool coop_fwd128 = true;
    bool gen_mfma2 = true;                        // ... with 128 x 128 block tiles and LDS-shared operand panels
    bool gen_mfma = true;                         // generic path: MFMA split-K weight gradient for wide layers
    bool coop128 = true;                          // H = 128: use the cooperative reverse kernel (else global-atomic slabs)
    int64_t coop_fwd_max_tiles = 0;               // forward: cooperative kernel for batches up to this many tiles
    int64_t stage_min_tiles = 0;                  // batches with fewer 16-point tiles use the unstaged kernels (latency-bound regime)
    hipStream_t side = nullptr;                   // boundary batch runs here, concurrently with the collocation batch
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    float *gslab_bc = nullptr, *grad_bc = nullptr;
    bool bc_inflight = false;
    // gpe_run replays one captured step (hipGraph) while every by-value launch argument is unchanged
    bool use_graph = false;
    int graph_steps = 8;           // steps captured per graph (GPE_GRAPH_STEPS)
    bool graph_forced = false;     // GPE_GRAPH set: at every batch size; otherwise only up to graph_max_points
    int64_t graph_max_points = 16384;
    hipStream_t cap_stream = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    std::vector<char> graph_key;
    bool prof = false;
    std::vector<hipEvent_t> ev_pool;      // pairs: [2i] start, [2i+1] stop
    std::vector<int> ev_kind;             // 0 forward, 1 reverse
    size_t ev_used = 0;
    const float** orth_dev = nullptr;
    const float* orth_host[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [4..6]: precomputed base
    Batch main, bc, sym, aux, mse;
    const float* bc_target = nullptr;
    // what the caller bound: collocation points / potential, boundary points.  When the boundary batch is small it is MERGED:
    // appended to the collocation batch (main.pts.b) and handled by the same launches; e->bc stays empty then.
    const float* ux = nullptr; const float* uV = nullptr; const float* uxb = nullptr;
    int64_t n_pde = 0, nb_user = 0, nb_merged = 0;
    bool merge_bc = true;
    const float* mse_target = nullptr;
    int num_cu = 256;
    int head_wg_per_cu = 1;        // head / seed kernels: workgroups per CU (each ends in one double atomic per global sum: ~22 ns apiece at one address)
    int head_threads = 1024;       // ... and threads per workgroup (GPE_HEAD_THREADS; 256 x 2 per CU until round 3: half the loads in flight, twice the atomics)
    // ---- data-parallel exchange inside the engine: RCCL on a dedicated stream (gpe_comm_init) ----
    struct Rccl {
        void* dl = nullptr;
        ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
        ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
        ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
        ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
        const char* (*GetErrorString)(ncclResult_t) = nullptr;
    } rccl;
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr;      // compute -> exchange, exchange -> compute
    std::vector<hipEvent_t> ev_bucket;                // one per linear map: its gradient slice is final
    // opt-in one-step-stale gradient (gpe_comm_set_async): the all-reduce of g_t runs on the exchange stream behind the forward of
    // step t+1, whose update applies g_t; changes the trajectory -- never on by default
    bool async_grad = false;
    float* grad_alt = nullptr;                        // second gradient buffer (P + GT_COUNT)
    double* dbl_prev = nullptr;                       // sums of the step whose gradient is in flight
    hipEvent_t ev_g[2] = {nullptr, nullptr};
    int64_t async_t = 0;
    bool dp_bucket = false;                           // set by gpe_step_dp: the generic reverse pass hands finished layers to the comm stream
    int64_t dp_collectives = 0;                       // all-reduces issued since gpe_comm_init (bench / tests)
    bool dp_inline = true;                            // synchronous step on the fused / wide sets: collectives on the compute stream (GPE_DP_INLINE=0: exchange stream)
    int phase = 0;                 // 0 idle, 1 after begin, 2 after backward
    std::string err;
    double* sums() { return dbl; }
    double* lsums() { return dbl + S_COUNT; }
    double* dsc() { return dbl + S_COUNT + LS_COUNT; }
};

static void free_batch(Batch& b) {
    for (void* p : b.allocs) (void)hipFree(p);
    b = Batch();
}

template <typename T>
static int dev_alloc(gpe_engine* e, Batch* b, T** out, size_t count) {
    void* p = nullptr;
    HIPCHK(e, hipMalloc(&p, count * sizeof(T) + 256));
    HIPCHK(e, hipMemsetAsync(p, 0, count * sizeof(T) + 256, e->stream));
    if (b) b->allocs.push_back(p);
    *out = (T*)p;
    return GPE_OK;
}

static int setup_batch(gpe_engine* e, Batch& b, const float* x, int64_t n, int C, int E, bool with_head, const float* V,
                       bool with_store = true) {
    if (b.n == n && b.C == C && b.E == E && b.O) { b.pts = Pts{x, nullptr, n}; b.V = V; return GPE_OK; }
    HIPCHK(e, hipStreamSynchronize(e->stream));
    float* keep_xown = nullptr;
    (void)keep_xown;
    free_batch(b);
    b.pts = Pts{x, nullptr, n}; b.V = V; b.n = n; b.C = C; b.E = E;
    b.ld = round_up(n, 64);
    const int no = e->nd.n_out;
    int rc;
    if ((rc = dev_alloc(e, &b, &b.O, (size_t)C * no * b.ld))) return rc;
    if ((rc = dev_alloc(e, &b, &b.Ob, (size_t)C * no * b.ld))) return rc;
    if (with_head) {
        if ((rc = dev_alloc(e, &b, &b.u, (size_t)no * b.ld))) return rc;
        if ((rc = dev_alloc(e, &b, &b.Hu, (size_t)no * b.ld))) return rc;
        if ((rc = dev_alloc(e, &b, &b.ux, (size_t)no * e->nd.dim * b.ld))) return rc;      // [n_out][dim][ld]
    }
    const int L = e->nd.n_lin - 1;
    if (e->path == GPE_PATH_FUSED) {
        int64_t ntiles = (n + 15) / 16;
        size_t cnt = with_store ? (size_t)ntiles * (L - 1) * C * e->H * 16 : 0;     // forward-only batches keep nothing
        if ((rc = dev_alloc(e, &b, &b.stored, cnt ? cnt : 4))) return rc;
        if (e->wide && with_store) {
            const size_t zc = (size_t)ntiles * C * e->H * 16;
            if ((rc = dev_alloc(e, &b, &b.Z0, zc))) return rc;
            if ((rc = dev_alloc(e, &b, &b.Z1, zc))) return rc;
        }
    } else {
        int maxW = 1;
        for (int h = 0; h < L; ++h) {
            float* s;
            if ((rc = dev_alloc(e, &b, &s, (size_t)C * e->nd.width[h + 1] * b.ld))) return rc;
            b.S.push_back(s);
            if (e->nd.width[h + 1] > maxW) maxW = e->nd.width[h + 1];
        }
        if ((rc = dev_alloc(e, &b, &b.A0, (size_t)C * maxW * b.ld))) return rc;
        if ((rc = dev_alloc(e, &b, &b.A1, (size_t)C * maxW * b.ld))) return rc;
        if (e->cfg.net_kind == GPE_NET_RESIDUAL && (rc = dev_alloc(e, &b, &b.A2, (size_t)C * maxW * b.ld))) return rc;
    }
    return GPE_OK;
}

// ---- MLP forward / backward dispatch ---------------------------------------------------------------
// (C, E) pairs that exist: value only (1,0); 1D (3,1); Laplacian-channel training batches 2D (4,1), 3D (5,1); full diagonal
// second derivatives for gpe_forward_jets 2D (5,2), 3D (7,3) -- forward kernels only.
#define CE_CASE(Cc, Ee, ...) case (Cc) * 10 + (Ee): { constexpr int CC = Cc; constexpr int EE = Ee; __VA_ARGS__; } break;
#ifdef GPE_FAST_BUILD    // kernel-tuning builds (tools/build_variant.sh): only what the 2D n_out = 1, H = 64 workloads launch
#define DISPATCH_TRAIN(bb, ...)                                        \
    switch ((bb).C * 10 + (bb).E) {                                    \
        CE_CASE(1, 0, __VA_ARGS__) CE_CASE(4, 1, __VA_ARGS__) CE_CASE(5, 1, __VA_ARGS__)         \
        default: FAIL(e, GPE_ERR_INVALID, "fast build: channels (%d,%d) not compiled", (bb).C, (bb).E); \
    }
#define DISPATCH_FWD(bb, ...) DISPATCH_TRAIN(bb, __VA_ARGS__)
#else
#define DISPATCH_TRAIN(bb, ...)                                        \
    switch ((bb).C * 10 + (bb).E) {                                    \
        CE_CASE(1, 0, __VA_ARGS__) CE_CASE(3, 1, __VA_ARGS__) CE_CASE(4, 1, __VA_ARGS__) CE_CASE(5, 1, __VA_ARGS__) \
        default: FAIL(e, GPE_ERR_INVALID, "bad channel pair (%d,%d)", (bb).C, (bb).E); \
    }
#define DISPATCH_FWD(bb, ...)                                          \
    switch ((bb).C * 10 + (bb).E) {                                    \
        CE_CASE(1, 0, __VA_ARGS__) CE_CASE(3, 1, __VA_ARGS__) CE_CASE(4, 1, __VA_ARGS__) CE_CASE(5, 1, __VA_ARGS__) \
        CE_CASE(5, 2, __VA_ARGS__) CE_CASE(7, 3, __VA_ARGS__)          \
        default: FAIL(e, GPE_ERR_INVALID, "bad channel pair (%d,%d)", (bb).C, (bb).E); \
    }
#endif

#ifdef GPE_FAST_BUILD
#define F_LAUNCH(KERNEL, HH, CC, EE, WL, GRID, BLOCK, LDS, ...) \
    hipLaunchKernelGGL((KERNEL<HH, CC, EE, 1, WL>), dim3(GRID), dim3(BLOCK), LDS, e->stream, __VA_ARGS__)
#define B_LAUNCH(HH, CC, EE, WL, NH, GRID, BLOCK, LDS, ...) \
    hipLaunchKernelGGL((f_backward<HH, CC, EE, 1, WL, NH>), dim3(GRID), dim3(BLOCK), LDS, e->stream, __VA_ARGS__)
#else
#define F_LAUNCH(KERNEL, HH, CC, EE, WL, GRID, BLOCK, LDS, ...)                                                    \
    do {                                                                                                         \
        if (e->nd.n_out == 1) hipLaunchKernelGGL((KERNEL<HH, CC, EE, 1, WL>), dim3(GRID), dim3(BLOCK), LDS, e->stream, __VA_ARGS__); \
        else                  hipLaunchKernelGGL((KERNEL<HH, CC, EE, 2, WL>), dim3(GRID), dim3(BLOCK), LDS, e->stream, __VA_ARGS__); \
    } while (0)
#define B_LAUNCH(HH, CC, EE, WL, NH, GRID, BLOCK, LDS, ...)                                                        \
    do {                                                                                                         \
        if (e->nd.n_out == 1) hipLaunchKernelGGL((f_backward<HH, CC, EE, 1, WL, NH>), dim3(GRID), dim3(BLOCK), LDS, e->stream, __VA_ARGS__); \
        else                  hipLaunchKernelGGL((f_backward<HH, CC, EE, 2, WL, NH>), dim3(GRID), dim3(BLOCK), LDS, e->stream, __VA_ARGS__); \
    } while (0)
#endif

static size_t fused_w_bytes(gpe_engine* e) { return (size_t)(e->nd.n_lin - 2) * e->H * e->H * sizeof(float); }
static size_t fused_small_bytes(gpe_engine* e) {       // = small_count() of gpe_fused.h, rounded up to 16 bytes
    size_t n = (size_t)(4 + (e->nd.n_lin - 2) + e->nd.n_out) * e->H + 4;
    return ((n + 3) & ~(size_t)3) * sizeof(float);
}
static bool staged_batch(gpe_engine* e, const Batch& b) { return (b.n + 15) / 16 >= e->stage_min_tiles; }
static size_t fused_fwd_lds(gpe_engine* e, bool staged) { return fused_small_bytes(e) + (e->fwd_wlds && staged ? fused_w_bytes(e) : 0); }

static bool coop_shape(gpe_engine* e);
static unsigned fused_grid(gpe_engine* e, int64_t n, int waves_per_block, int blocks_per_cu);
static bool fwd_coop(gpe_engine* e, const Batch& b) {
    if (!coop_shape(e) || e->coop == 0) return false;
    if (e->H == 128) return e->coop_fwd128;           // wide layers: the cooperative forward wins at every size (measured)
    if (e->nd.n_lin - 2 > 3 || e->cfg.net_kind == GPE_NET_RESIDUAL) return true;    // four / five maps at H <= 64, residual blocks: only the cooperative kernels take them
    return (b.n + 15) / 16 <= e->coop_fwd_max_tiles;
}
#define HEAD_SLOTS 512
static bool seed_in_reverse(gpe_engine* e);
// no Riesz term, Rayleigh-quotient eigenvalue, no regularisers: the loss the fused head / seed code paths implement
static bool plain_terms(const gpe_engine* e) {
    return e->cfg.w_riesz == 0.f && e->cfg.lambda_kind == GPE_LAMBDA_RAYLEIGH && e->cfg.w_reg_f == 0.f && e->cfg.w_reg_lam == 0.f;
}
// whole steps (gpe_step / gpe_run) of the small-batch class whose reverse kernel forms the seeds: the cooperative forward kernel runs
// the head too, k_head_pde is not launched and the step sums are added in a fixed order
// problem class whose head the forward kernels can run (head_point_real): real psi, no orthogonality / Riesz / symmetry terms
static bool head_class(gpe_engine* e) {
    return e->fuse_head && e->head_slots && e->path == GPE_PATH_FUSED && !e->wide && e->H <= 64 && e->nd.n_out == 1 && !e->cfg.complex_psi &&
           e->ph.n_orth == 0 && plain_terms(e) && e->cfg.w_sym == 0.f && e->main.C >= 3 && e->main.n > 0;
}
// ... by the cooperative forward kernel (small batches; the reverse kernel forms the seeds and adds the triples)
static bool head_fusable_coop(gpe_engine* e) {
    // (four / five maps and residual blocks have no seed-forming reverse kernel: k_seed_pde adds their triples, as it does for large batches)
    const bool deep = e->nd.n_lin - 2 > 3 || e->cfg.net_kind == GPE_NET_RESIDUAL;
    return head_class(e) && e->main.n <= e->fuse_head_max && fwd_coop(e, e->main) && (seed_in_reverse(e) || deep) && fused_grid(e, e->main.n, 1, 2) <= HEAD_SLOTS;
}
// ... by the per-wave-tile forward kernel (large batches; k_seed_pde, or the seed-forming reverse kernel, adds the triples)
static bool head_fusable_tile(gpe_engine* e) {
    return head_class(e) && !fwd_coop(e, e->main) && !e->fwd_b6 && e->main.n >= e->fuse_head_tile_min && fused_grid(e, e->main.n, 4, e->fwd_wg_per_cu) <= HEAD_SLOTS;
}
static bool head_fusable(gpe_engine* e) { return head_fusable_coop(e) || head_fusable_tile(e); }
static bool head_in_forward(gpe_engine* e) { return e->fuse_head && e->fh_want && head_fusable(e); }
template <int HH, int CC, int EE, int NO>
static void launch_fcoop_no(gpe_engine* e, Batch& b, unsigned grid, size_t lds, int store) {
#define CARGS e->nd, e->theta, e->Wpk, b.pts, b.stored, b.O, b.n, b.ld, store
    if constexpr (HH <= 64 && NO == 1 && CC >= 3) {
        if (e->fh_now && &b == &e->main) {
            const HeadArgs ha{e->ph, e->base_norm, b.V, (const float* const*)e->orth_dev, e->bc_target, b.u, b.Hu, b.Ob, e->n_pde, b.ld, e->head_slots};
            const size_t ldsh = lds + (size_t)CC * 16 * sizeof(float);
            if (e->cfg.net_kind == GPE_NET_RESIDUAL) {
                if (e->nd.n_lin - 2 == 2) hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, 1, 2, true, true>), dim3(grid), dim3(HH * 4), ldsh, e->stream, CARGS, ha);
                else hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, 1, 4, true, true>), dim3(grid), dim3(HH * 4), ldsh, e->stream, CARGS, ha);
                return;
            }
            switch (e->nd.n_lin - 2) {
                case 1: hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, 1, 1, true>), dim3(grid), dim3(HH * 4), ldsh, e->stream, CARGS, ha); break;
                case 2: hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, 1, 2, true>), dim3(grid), dim3(HH * 4), ldsh, e->stream, CARGS, ha); break;
                case 3: hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, 1, 3, true>), dim3(grid), dim3(HH * 4), ldsh, e->stream, CARGS, ha); break;
                case 4: hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, 1, 4, true>), dim3(grid), dim3(HH * 4), ldsh, e->stream, CARGS, ha); break;
                default: hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, 1, 5, true>), dim3(grid), dim3(HH * 4), ldsh, e->stream, CARGS, ha); break;
            }
            return;
        }
    }
#undef CARGS
    const HeadArgs nohead{};
#define CARGS e->nd, e->theta, e->Wpk, b.pts, b.stored, b.O, b.n, b.ld, store, nohead
    if constexpr (HH <= 64 && NO == 1) {
        if (e->cfg.net_kind == GPE_NET_RESIDUAL) {        // one or two residual blocks (checked at gpe_create)
            if (e->nd.n_lin - 2 == 2) hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, 1, 2, false, true>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS);
            else hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, 1, 4, false, true>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS);
            return;
        }
    }
    switch (e->nd.n_lin - 2) {
        case 1: hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, NO, 1>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS); break;
        case 2: hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, NO, 2>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS); break;
        case 3: hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, NO, 3>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS); break;
        case 4:
            if constexpr (HH == 128 || NO == 1)
                hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, NO, 4>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS);
            break;
        default:
            if constexpr (HH == 128 || NO == 1)
                hipLaunchKernelGGL((f_forward_coop<HH, CC, EE, NO, 5>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS);
            break;
    }
#undef CARGS
}
template <int HH, int CC, int EE>
static void launch_f_forward(gpe_engine* e, Batch& b, unsigned grid, int store) {
    if constexpr (HH <= 64 || (HH == 128 && CC <= 4)) {
        if (fwd_coop(e, b)) {
            const int NT = HH / 16;
            const size_t lds = fused_small_bytes(e) + ((size_t)2 * CC * NT * 256 + (size_t)NT * e->nd.n_out * CC * 16) * sizeof(float);
            const unsigned g = fused_grid(e, b.n, 1, HH == 128 ? 1 : 2);
#ifdef GPE_FAST_BUILD
            launch_fcoop_no<HH, CC, EE, 1>(e, b, g, lds, store);
#else
            if (e->nd.n_out == 1) launch_fcoop_no<HH, CC, EE, 1>(e, b, g, lds, store);
            else launch_fcoop_no<HH, CC, EE, 2>(e, b, g, lds, store);
#endif
            return;
        }
    }
    // large batches on two workgroups per CU: uneven split of each CU's tiles between its two workgroups (GPE_FWD_SHARE, / 1024)
    const int fshare = (e->fwd_share > 0 && grid == (unsigned)(2 * e->num_cu) && (b.n + 15) / 16 >= 4 * e->share_min_tiles * (int64_t)grid) ? e->fwd_share : 0;
    const HeadArgs nohead{};
    if constexpr (HH > 64) {
        F_LAUNCH(f_forward, HH, CC, EE, false, grid, 256, fused_fwd_lds(e, false), e->nd, e->theta, e->Wpk, b.pts, b.stored, b.O, b.n, b.ld, store, fshare, nohead);
        return;
    }
    if constexpr (HH <= 64) {
        if (e->fwd_b6 && staged_batch(e, b)) {
            // weights (three bf16 pieces, 6 bytes each) in LDS when two workgroups per CU still fit
            const size_t w6 = (size_t)(e->nd.n_lin - 2) * HH * HH * 6;
            const bool wl = 2 * (fused_small_bytes(e) + w6) <= (size_t)160 * 1024;
            const size_t lds6 = fused_small_bytes(e) + (wl ? w6 : 0);
#define B6ARGS e->nd, e->theta, e->WpkT, b.pts, b.stored, b.O, b.n, b.ld, store
#ifdef GPE_FAST_BUILD
            if (wl) hipLaunchKernelGGL((f_forward_b6<HH, CC, EE, 1, true>), dim3(grid), dim3(256), lds6, e->stream, B6ARGS);
            else hipLaunchKernelGGL((f_forward_b6<HH, CC, EE, 1, false>), dim3(grid), dim3(256), lds6, e->stream, B6ARGS);
#else
            if (e->nd.n_out == 1) {
                if (wl) hipLaunchKernelGGL((f_forward_b6<HH, CC, EE, 1, true>), dim3(grid), dim3(256), lds6, e->stream, B6ARGS);
                else hipLaunchKernelGGL((f_forward_b6<HH, CC, EE, 1, false>), dim3(grid), dim3(256), lds6, e->stream, B6ARGS);
            } else {
                if (wl) hipLaunchKernelGGL((f_forward_b6<HH, CC, EE, 2, true>), dim3(grid), dim3(256), lds6, e->stream, B6ARGS);
                else hipLaunchKernelGGL((f_forward_b6<HH, CC, EE, 2, false>), dim3(grid), dim3(256), lds6, e->stream, B6ARGS);
            }
#endif
#undef B6ARGS
            return;
        }
    }
    if constexpr (HH <= 64 && CC >= 3) {
        if (e->fh_now && &b == &e->main && e->nd.n_out == 1) {     // whole step, large batch: the head rides in this kernel
            const HeadArgs ha{e->ph, e->base_norm, b.V, (const float* const*)e->orth_dev, e->bc_target, b.u, b.Hu, b.Ob, e->n_pde, b.ld, e->head_slots};
            if (e->fwd_wlds && staged_batch(e, b))
                hipLaunchKernelGGL((f_forward<HH, CC, EE, 1, true, true>), dim3(grid), dim3(256), fused_fwd_lds(e, true), e->stream, e->nd, e->theta, e->Wpk,
                                   b.pts, b.stored, b.O, b.n, b.ld, store, fshare, ha);
            else
                hipLaunchKernelGGL((f_forward<HH, CC, EE, 1, false, true>), dim3(grid), dim3(256), fused_fwd_lds(e, false), e->stream, e->nd, e->theta, e->Wpk,
                                   b.pts, b.stored, b.O, b.n, b.ld, store, fshare, ha);
            return;
        }
    }
    if (e->fwd_wlds && staged_batch(e, b))
        F_LAUNCH(f_forward, HH, CC, EE, true, grid, 256, fused_fwd_lds(e, true), e->nd, e->theta, e->Wpk, b.pts, b.stored, b.O, b.n, b.ld, store, fshare, nohead);
    else
        F_LAUNCH(f_forward, HH, CC, EE, false, grid, 256, fused_fwd_lds(e, false), e->nd, e->theta, e->Wpk, b.pts, b.stored, b.O, b.n, b.ld, store, fshare, nohead);
}
// reverse-kernel variant for one batch: 3 = cooperative (a workgroup per tile, a wave per 16-feature slice),
// 2 = weight gradients in registers (1 wave/SIMD),
// 0 = plain (LDS-atomic gradients, weights from L2; also the fastest when every wave sees only a tile or two)
static bool coop_shape(gpe_engine* e) {
    if (e->path != GPE_PATH_FUSED) return false;
    const int maps = e->nd.n_lin - 2;                 // hidden -> hidden maps
    // H <= 64: one to three maps on the pipelined / cooperative kernels; round 4: FOUR and FIVE maps (real psi) on the cooperative ones too -- the
    // per-wave-tile kernels cannot hold a fourth map's weight gradient in registers, nor its weights in 64 KB of LDS, and ran such networks 2.4-3x
    // slower per FLOP (profiles/r04/deep_h64_networks.txt)
    if (e->H <= 64) return maps >= 1 && maps <= (e->nd.n_out == 1 ? 5 : 3);
    return e->H == 128 && maps >= 1 && maps <= 5 && e->nd.dim <= 2 && e->coop128;    // 8 waves per workgroup, weights streamed from L2
}
static int bwd_kind(gpe_engine* e, const Batch& b);
static bool use_pipe(gpe_engine* e, int C);
// small batches of the common problem class (real psi, no orthogonality / Riesz / symmetry terms) on the pipelined reverse kernel:
// the kernel forms the seeds itself and k_seed_pde is not launched
static bool seed_in_reverse(gpe_engine* e) {
    return e->fuse_seed && e->path == GPE_PATH_FUSED && !e->wide && e->nd.n_out == 1 && !e->cfg.complex_psi && e->ph.n_orth == 0 &&
           plain_terms(e) && e->cfg.w_sym == 0.f && e->main.C >= 3 && e->main.n > 0 && e->n_pde <= e->fuse_seed_max &&
           use_pipe(e, e->main.C) && bwd_kind(e, e->main) == 3;
}
// H = 128 in 1D / 2D has two reverse passes behind the cooperative forward kernel (same stored-activation format): one launch per map
// (w_bwd_map: no spills, -3.6 % at 4 096 tiles) and the single cooperative launch (f_backward_coop<128>: fewer launches, -10 % at 256
// tiles, -4 % at 1 024; profiles/r04/wide_small_batch_ab.txt).  The per-map form from wide_min_tiles tiles on (GPE_WIDE_MIN_TILES).
static bool wide_reverse(gpe_engine* e, const Batch& b);
static int bwd_kind(gpe_engine* e, const Batch& b) {
    if (coop_shape(e) && e->coop != 0 && (e->coop == 1 || (b.n + 15) / 16 <= e->coop_max_tiles || (e->H <= 64 && (e->nd.n_lin - 2 > 3 || e->cfg.net_kind == GPE_NET_RESIDUAL)))) return 3;
    if (e->H > 64 || !staged_batch(e, b)) return 0;
    return e->bwd_racc ? 2 : 0;
}
static bool wide_reverse(gpe_engine* e, const Batch& b) {
    if (!e->wide) return false;
    if (e->wide_fwd || e->H != 128 || e->nd.dim > 2) return true;              // H = 256, 3D: the wide set is the only one
    return (b.n + 15) / 16 >= e->wide_min_tiles || bwd_kind(e, b) != 3;
}
// f_backward_pipe: two z and two X^T exchange buffers; taken when two workgroups per CU still fit
static size_t pipe_lds(gpe_engine* e, int C) {
    const int H = e->H, NT = H / 16, L = e->nd.n_lin - 1;
    const size_t n_gsm = ((size_t)(L - 1 + e->nd.n_out) * H + 4 + 3) & ~(size_t)3;
    return (n_gsm + 4 * (size_t)H) * sizeof(float) + fused_small_bytes(e) + (size_t)4 * C * NT * 256 * sizeof(float);
}
static bool use_pipe(gpe_engine* e, int C) {
    return e->bwd_pipe && !e->bwd_b6 && e->H <= 64 && e->cfg.net_kind == GPE_NET_MLP && e->nd.n_lin - 2 >= 1 && e->nd.n_lin - 2 <= 3 && 2 * (pipe_lds(e, C) + 1024) <= (size_t)160 * 1024;
}
static size_t coop_lds(gpe_engine* e, int C) {
    if (use_pipe(e, C)) return pipe_lds(e, C);
    const int H = e->H, NT = H / 16, L = e->nd.n_lin - 1;
    const size_t n_gsm = ((size_t)(L - 1 + e->nd.n_out) * H + 4 + 3) & ~(size_t)3;
    const size_t zb = e->bwd_b6 && H <= 64 ? (size_t)3 * C * (NT / 2) * 256 : (size_t)C * NT * 256;     // B6: three bf16 pieces, 24 B per four values
    return (n_gsm + 4 * (size_t)H) * sizeof(float) + fused_small_bytes(e) + (zb + 2 * (size_t)C * NT * F_TILE) * sizeof(float);
}
template <int HH, int CC, int EE, int NO>
static void set_pipe_lds(int bytes) {
    if constexpr (!(HH == 64 && CC == 5)) {
        (void)hipFuncSetAttribute((const void*)f_backward_pipe<HH, CC, EE, NO, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void*)f_backward_pipe<HH, CC, EE, NO, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void*)f_backward_pipe<HH, CC, EE, NO, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if constexpr (NO == 1 && CC >= 3) {
            (void)hipFuncSetAttribute((const void*)f_backward_pipe<HH, CC, EE, 1, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            (void)hipFuncSetAttribute((const void*)f_backward_pipe<HH, CC, EE, 1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            (void)hipFuncSetAttribute((const void*)f_backward_pipe<HH, CC, EE, 1, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        }
    }
}
template <int HH, int CC, int EE, int NO>
static void launch_coop_no(gpe_engine* e, Batch& b, unsigned grid, size_t lds) {
#define CARGS e->nd, e->theta, e->WpkT, b.pts, b.stored, b.Ob, e->gslab, b.n, b.ld, e->Ppad
    if constexpr (HH <= 64) {
        if (e->bwd_b6 && e->nd.n_lin - 2 <= 3 && e->cfg.net_kind == GPE_NET_MLP) {
            switch (e->nd.n_lin - 2) {
                case 1: hipLaunchKernelGGL((f_backward_coop<HH, CC, EE, NO, 1, true>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS); break;
                case 2: hipLaunchKernelGGL((f_backward_coop<HH, CC, EE, NO, 2, true>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS); break;
                default: hipLaunchKernelGGL((f_backward_coop<HH, CC, EE, NO, 3, true>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS); break;
            }
            return;
        }
        if constexpr (!(HH == 64 && CC == 5)) if (use_pipe(e, CC)) {       // (H = 64 in 3D: two workgroups' exchange buffers exceed the LDS)
            if constexpr (NO == 1 && CC >= 3) {
                if (e->seedf_now && &b == &e->main) {       // small batch: the kernel forms the seeds itself (k_seed_pde was not launched)
                    const SeedArgs sa{e->ph, b.V, b.u, b.Hu, (const double*)e->sums(), e->dsc(), e->n_pde,
                                      e->fh_now ? (const double*)e->head_slots : nullptr, e->fh_now ? e->fh_nslots : 0,
                                      e->sums(), e->lsums(), 0};
                    switch (e->nd.n_lin - 2) {
                        case 1: hipLaunchKernelGGL((f_backward_pipe<HH, CC, EE, 1, 1, true>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS, sa); break;
                        case 2: hipLaunchKernelGGL((f_backward_pipe<HH, CC, EE, 1, 2, true>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS, sa); break;
                        default: hipLaunchKernelGGL((f_backward_pipe<HH, CC, EE, 1, 3, true>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS, sa); break;
                    }
                    return;
                }
            }
            SeedArgs none{};
            // large batches on two workgroups per CU: uneven split of each CU's tiles between its two workgroups (GPE_PIPE_SHARE, / 1024)
            if (e->pipe_share > 0 && grid == (unsigned)(2 * e->num_cu) && (b.n + 15) / 16 >= e->share_min_tiles * (int64_t)grid) none.old_share_q10 = e->pipe_share;
            switch (e->nd.n_lin - 2) {
                case 1: hipLaunchKernelGGL((f_backward_pipe<HH, CC, EE, NO, 1>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS, none); break;
                case 2: hipLaunchKernelGGL((f_backward_pipe<HH, CC, EE, NO, 2>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS, none); break;
                default: hipLaunchKernelGGL((f_backward_pipe<HH, CC, EE, NO, 3>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS, none); break;
            }
            return;
        }
    }
    if constexpr (HH <= 64 && NO == 1) {
        if (e->cfg.net_kind == GPE_NET_RESIDUAL) {
            if (e->nd.n_lin - 2 == 2) hipLaunchKernelGGL((f_backward_coop<HH, CC, EE, 1, 2, false, true>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS);
            else hipLaunchKernelGGL((f_backward_coop<HH, CC, EE, 1, 4, false, true>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS);
            return;
        }
    }
    switch (e->nd.n_lin - 2) {
        case 1: hipLaunchKernelGGL((f_backward_coop<HH, CC, EE, NO, 1>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS); break;
        case 2: hipLaunchKernelGGL((f_backward_coop<HH, CC, EE, NO, 2>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS); break;
        case 3: hipLaunchKernelGGL((f_backward_coop<HH, CC, EE, NO, 3>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS); break;
        case 4:
            if constexpr (HH == 128 || NO == 1)
                hipLaunchKernelGGL((f_backward_coop<HH, CC, EE, NO, 4>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS);
            break;
        default:
            if constexpr (HH == 128 || NO == 1)
                hipLaunchKernelGGL((f_backward_coop<HH, CC, EE, NO, 5>), dim3(grid), dim3(HH * 4), lds, e->stream, CARGS);
            break;
    }
#undef CARGS
}
template <int HH, int CC, int EE>
static void launch_f_backward_coop(gpe_engine* e, Batch& b, unsigned grid, size_t lds) {
    {
#ifdef GPE_FAST_BUILD
        launch_coop_no<HH, CC, EE, 1>(e, b, grid, lds);
#else
        if (e->nd.n_out == 1) launch_coop_no<HH, CC, EE, 1>(e, b, grid, lds);
        else launch_coop_no<HH, CC, EE, 2>(e, b, grid, lds);
#endif
    }
}
template <int HH, int CC, int EE>
static void launch_f_backward(gpe_engine* e, Batch& b, unsigned grid, size_t lds) {
    if constexpr (HH > 64) {
        B_LAUNCH(HH, CC, EE, false, 0, grid, 256, lds, e->nd, e->theta, e->WpkT, b.pts, b.stored, b.Ob, e->gslab, b.n, b.ld, e->Ppad, e->nslab_g);
        return;
    } else {
#define BARGS e->nd, e->theta, e->WpkT, b.pts, b.stored, b.Ob, e->gslab, b.n, b.ld, e->Ppad, e->nslab_g
    const int kind = bwd_kind(e, b);
    if (kind == 3) launch_f_backward_coop<HH, CC, EE>(e, b, grid, lds);
    else if (kind == 2) {
        switch (e->nd.n_lin - 2) {
            case 1: B_LAUNCH(HH, CC, EE, true, 1, grid, 256, lds, BARGS); break;
            case 2: B_LAUNCH(HH, CC, EE, true, 2, grid, 256, lds, BARGS); break;
            default: B_LAUNCH(HH, CC, EE, true, 3, grid, 256, lds, BARGS); break;
        }
    }
    else
        B_LAUNCH(HH, CC, EE, false, 0, grid, 256, lds, BARGS);
#undef BARGS
    }
}

// persistent grids: forward 256-thread blocks, 2 per CU; reverse 256-thread x 2 per CU, or 512-thread x 1 per CU (WLDS)
static unsigned fused_grid(gpe_engine* e, int64_t n, int waves_per_block, int blocks_per_cu) {
    int64_t ntiles = (n + 15) / 16;
    int64_t blocks = (ntiles + waves_per_block - 1) / waves_per_block;
    int64_t cap = (int64_t)e->num_cu * blocks_per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

static size_t fused_bwd_lds(gpe_engine* e, int C, int kind) {
    if (e->H > 64) return ((size_t)4 * e->H + (size_t)4 * C * F_TILE) * sizeof(float) + fused_small_bytes(e);
    const int nwaves = 4;
    const bool w = kind != 0;
    return ((size_t)e->Ppad + 4 * (size_t)e->H + (size_t)nwaves * C * F_TILE) * sizeof(float) + fused_small_bytes(e) +
           (w ? fused_w_bytes(e) : 0);
}

static int ensure_packed(gpe_engine* e) {
    if (e->path != GPE_PATH_FUSED || !e->packed_dirty) return GPE_OK;
    const int L = e->nd.n_lin - 1;
    int total = (L - 1) * e->H * e->H;
    hipLaunchKernelGGL(k_pack_weights, dim3(cdiv(total, 256)), dim3(256), 0, e->stream, e->nd, e->H, e->theta, e->Wpk,
                       e->WpkT);
    HIPCHK(e, hipGetLastError());
    e->packed_dirty = false;
    return GPE_OK;
}

// zero the step accumulators (+ repack weights when stale) in one launch
static int launch_begin(gpe_engine* e, bool force = false) {
    // fused path: k_update zeroed the sums and repacked the weights, and the first slab reduction assigns -> nothing to do
    if (!force && e->path == GPE_PATH_FUSED && e->acc_clean && !e->packed_dirty) { e->acc_clean = false; return GPE_OK; }
    e->acc_clean = false;
    int n_pack = 0;
    if (e->path == GPE_PATH_FUSED && e->packed_dirty) n_pack = (e->nd.n_lin - 2) * e->H * e->H;
    const int n_dbl = S_COUNT + LS_COUNT + 4, n_grad = e->P + GT_COUNT, n_bc = e->P;
    const int n = std::max(std::max(n_pack, n_dbl), n_grad);
    hipLaunchKernelGGL(k_begin, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, e->nd, e->H, e->theta, e->Wpk, e->WpkT, n_pack,
                       e->dbl, n_dbl, e->grad, n_grad, e->grad_bc, n_bc);
    HIPCHK(e, hipGetLastError());
    if (n_pack) e->packed_dirty = false;
    return GPE_OK;
}

static void prof_mark(gpe_engine* e, int kind, bool start) {
    if (!e->prof) return;
    if (start) {
        if (2 * e->ev_used + 1 >= e->ev_pool.size()) {
            hipEvent_t a, b2;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b2) != hipSuccess) return;
            e->ev_pool.push_back(a); e->ev_pool.push_back(b2); e->ev_kind.push_back(kind);
        } else e->ev_kind[e->ev_used] = kind;
        e->ev_used++;
    }
    if (e->ev_used == 0) return;
    size_t idx = e->ev_used - 1;
    (void)hipEventRecord(e->ev_pool[2 * idx + (start ? 0 : 1)], e->stream);
}

static WideCall wide_call(gpe_engine* e, Batch& b) {
    WideCall a;
    a.nd = e->nd; a.theta = e->theta; a.Wpk = e->Wpk; a.WpkT = e->WpkT; a.pts = b.pts; a.stored = b.stored; a.O = b.O; a.Ob = b.Ob;
    a.Z0 = b.Z0; a.Z1 = b.Z1; a.gslab = e->gslab; a.N = b.n; a.ld = b.ld; a.Ppad = e->Ppad; a.H = e->H; a.C = b.C; a.E = b.E;
    a.num_cu = e->num_cu; a.stream = e->stream;
    return a;
}

static int mlp_forward(gpe_engine* e, Batch& b, bool store) {
    if (b.n <= 0) return GPE_OK;
    const bool mark = e->prof && (&b == &e->main);
    if (e->path == GPE_PATH_FUSED) {
        int rc = ensure_packed(e);
        if (rc) return rc;
        unsigned grid = fused_grid(e, b.n, 4, e->fwd_wg_per_cu);
        if (mark) prof_mark(e, 0, true);
        if (e->wide_fwd) {
            const int wr = wide_forward(wide_call(e, b), store ? 1 : 0);
            if (wr < 0) FAIL(e, GPE_ERR_INVALID, "wide kernel set: channels (%d,%d) / n_out %d not compiled", b.C, b.E, e->nd.n_out);
            if (wr) FAIL(e, GPE_ERR_HIP, "w_forward launch: %s", hipGetErrorString((hipError_t)wr));
            if (mark) prof_mark(e, 0, false);
            return GPE_OK;
        }
#ifdef GPE_FAST_BUILD
        if (e->H != 64 || e->nd.n_out != 1) FAIL(e, GPE_ERR_INVALID, "fast build: only H = 64, n_out = 1 compiled");
        DISPATCH_FWD(b, launch_f_forward<64, CC, EE>(e, b, grid, store ? 1 : 0));
#else
        if (e->H == 128) {      // wide layers: one wave per SIMD, C <= 5 (checked at create)
            grid = fused_grid(e, b.n, 4, 1);
            switch (b.C * 10 + b.E) {     // wide layers: dim <= 2 (checked at create)
                case 10: launch_f_forward<128, 1, 0>(e, b, grid, store ? 1 : 0); break;
                case 31: launch_f_forward<128, 3, 1>(e, b, grid, store ? 1 : 0); break;
                case 41: launch_f_forward<128, 4, 1>(e, b, grid, store ? 1 : 0); break;
                case 52: launch_f_forward<128, 5, 2>(e, b, grid, store ? 1 : 0); break;
                default: FAIL(e, GPE_ERR_INVALID, "bad channel pair (%d,%d) for H = 128", b.C, b.E);
            }
        }
        else if (e->H == 64) { DISPATCH_FWD(b, launch_f_forward<64, CC, EE>(e, b, grid, store ? 1 : 0)); }
        else            { DISPATCH_FWD(b, launch_f_forward<32, CC, EE>(e, b, grid, store ? 1 : 0)); }
#endif
        if (mark) prof_mark(e, 0, false);
    } else {
        const NetDesc& nd = e->nd;
        for (int lin = 0; lin < nd.n_lin; ++lin) {
            const float* Sprev = lin > 0 ? b.S[lin - 1] : nullptr;
            float* Out = (lin == nd.n_lin - 1) ? b.O : b.S[lin];
            const float* Sskip = nd.skip[lin] >= 0 ? b.S[nd.skip[lin]] : nullptr;     // residual block: the VALU kernel adds the block input
            if (!Sskip && lin > 0 && nd.width[lin] % 64 == 0 && nd.width[lin + 1] % 256 == 0 && e->gen_mfma && e->gen_mfma2) {
                dim3 grid(cdiv(b.n, 16), nd.width[lin + 1] / 256);      // a block = one point tile x 256 outputs, jets shared through LDS
                DISPATCH_FWD(b, hipLaunchKernelGGL((g_fwd_layer_mfma2<CC, EE>), grid, dim3(256), 0, e->stream, nd, lin, e->theta,
                                                    Sprev, Out, b.n, b.ld));
            } else if (lin > 0 && nd.width[lin] % 64 == 0 && nd.width[lin + 1] % 64 == 0 && e->gen_mfma &&
                       (!Sskip || nd.width[nd.skip[lin] + 1] == nd.width[lin + 1])) {   // wide map: matrix cores (round 4: residual blocks too)
                dim3 grid(cdiv(cdiv(b.n, 16), 4), nd.width[lin + 1] / 64);
                DISPATCH_FWD(b, hipLaunchKernelGGL((g_fwd_layer_mfma<CC, EE>), grid, dim3(256), 0, e->stream, nd, lin, e->theta,
                                                    Sprev, Out, b.n, b.ld, Sskip));
            } else if (nd.width[lin + 1] >= 64) {      // wide layer: 16 output features per thread
                dim3 grid(cdiv(b.n, 256), cdiv(nd.width[lin + 1], G_FBW));
                DISPATCH_FWD(b, hipLaunchKernelGGL((g_fwd_layer<CC, EE, G_FBW>), grid, dim3(256), 0, e->stream, nd, lin, e->theta,
                                                    b.pts, Sprev, Out, b.n, b.ld, Sskip));
            } else {
                dim3 grid(cdiv(b.n, 256), cdiv(nd.width[lin + 1], G_FB));
                DISPATCH_FWD(b, hipLaunchKernelGGL((g_fwd_layer<CC, EE, G_FB>), grid, dim3(256), 0, e->stream, nd, lin, e->theta,
                                                    b.pts, Sprev, Out, b.n, b.ld, Sskip));
            }
        }
    }
    HIPCHK(e, hipGetLastError());
    return GPE_OK;
}

static int bc_join(gpe_engine* e);
static double bc_count(gpe_engine* e);
static int launch_tail(gpe_engine* e, bool add_bc);
static int dp_allreduce_after(gpe_engine* e, void* buf, size_t count, ncclDataType_t dt, hipEvent_t ev);
// Gradient of one batch into e->grad: assigned (first reverse pass of the step) or accumulated.  close: this is the last reverse pass of the step -- join the boundary batch's side stream, add its
// gradient and write the exchange tail (folded into the slab reduction on the fused path)
static int mlp_backward(gpe_engine* e, Batch& b, bool close = false, bool assign = true) {
    if (b.n <= 0) return close ? launch_tail(e, true) : GPE_OK;
    if (e->path == GPE_PATH_FUSED) {
        const int kind = bwd_kind(e, b);
        unsigned grid = kind == 3 ? fused_grid(e, b.n, 1, e->coop_wg_per_cu)
                                  : (kind == 2 ? fused_grid(e, b.n, 4, 1) : fused_grid(e, b.n, 4, 2));
        size_t lds = kind == 3 ? coop_lds(e, b.C) : fused_bwd_lds(e, b.C, kind);
        const bool mark = e->prof && (&b == &e->main);
        if (mark) prof_mark(e, 1, true);
        int nred = (int)grid;
        if (wide_reverse(e, b)) {
            if (!b.Z0) FAIL(e, GPE_ERR_STATE, "reverse pass on a forward-only batch");
            const int wr = wide_backward(wide_call(e, b));
            if (wr < 0) FAIL(e, GPE_ERR_INVALID, "wide kernel set: channels (%d,%d) / n_out %d not compiled", b.C, b.E, e->nd.n_out);
            if (wr) FAIL(e, GPE_ERR_HIP, "wide reverse launch: %s", hipGetErrorString((hipError_t)wr));
            nred = wide_groups(e->H, b.n, e->num_cu);
        } else {
#ifdef GPE_FAST_BUILD
        if (e->H != 64 || e->nd.n_out != 1) FAIL(e, GPE_ERR_INVALID, "fast build: only H = 64, n_out = 1 compiled");
        DISPATCH_TRAIN(b, launch_f_backward<64, CC, EE>(e, b, grid, lds));
#else
        if (e->H == 128 && kind == 3) {
            grid = fused_grid(e, b.n, 1, 1);
            nred = (int)grid;
            switch (b.C * 10 + b.E) {
                case 10: launch_f_backward_coop<128, 1, 0>(e, b, grid, lds); break;
                case 31: launch_f_backward_coop<128, 3, 1>(e, b, grid, lds); break;
                case 41: launch_f_backward_coop<128, 4, 1>(e, b, grid, lds); break;
                default: FAIL(e, GPE_ERR_INVALID, "bad channel pair (%d,%d) for H = 128", b.C, b.E);
            }
        }
        else if (e->H == 128) {
            grid = fused_grid(e, b.n, 4, 1);
            nred = e->nslab_g;
            HIPCHK(e, hipMemsetAsync(e->gslab, 0, (size_t)e->nslab_g * e->Ppad * sizeof(float), e->stream));
            switch (b.C * 10 + b.E) {
                case 10: launch_f_backward<128, 1, 0>(e, b, grid, lds); break;
                case 31: launch_f_backward<128, 3, 1>(e, b, grid, lds); break;
                case 41: launch_f_backward<128, 4, 1>(e, b, grid, lds); break;
                default: FAIL(e, GPE_ERR_INVALID, "bad channel pair (%d,%d) for H = 128", b.C, b.E);
            }
        }
        else if (e->H == 64) { DISPATCH_TRAIN(b, launch_f_backward<64, CC, EE>(e, b, grid, lds)); }
        else            { DISPATCH_TRAIN(b, launch_f_backward<32, CC, EE>(e, b, grid, lds)); }
#endif
        }
        if (mark) prof_mark(e, 1, false);
        HIPCHK(e, hipGetLastError());
        const float* add = nullptr;
        if (close) {
            int rc = bc_join(e);
            if (rc) return rc;
            if (e->bc_inflight) add = e->grad_bc;
            e->bc_inflight = false;
        }
        if (close && assign && e->fu_want && e->fuse_update && e->upd_ticket && !e->upd_snap && e->update_cache && e->P <= 13 * 1024 && !e->comm) {
            const int n_pack = (e->nd.n_lin - 2) * e->H * e->H;
            hipLaunchKernelGGL(k_reduce_update, dim3(cdiv(e->P, 64)), dim3(1024), 0, e->stream, e->gslab, nred, e->Ppad, e->P, e->grad, add,
                               (const double*)e->dsc(), e->upd_ticket, e->theta, e->am, e->av, (const double*)e->sums(), (const double*)e->lsums(),
                               e->ph, e->oc, e->od, e->hist, e->cap, e->last, bc_count(e), e->nd, e->H, e->Wpk, e->WpkT, n_pack, e->dbl,
                               (int)(S_COUNT + LS_COUNT + 4), ((e->fwd_b6 || e->bwd_b6 || e->H > 64) ? 1 : 2));
            e->fu_done = true;
        } else if (close && assign && e->fu_want && e->split_update && e->upd_snap_small && !e->upd_snap && !e->comm && !e->ext_exchange &&
                   !e->fwd_b6 && !e->bwd_b6 && e->H <= 64) {
            hipLaunchKernelGGL(k_grad_reduce_part, dim3(UPD_G), dim3(1024), 0, e->stream, e->gslab, nred, e->Ppad, e->P, e->grad, add,
                               (const double*)e->dsc(), (const double*)e->sums(), (const double*)e->lsums(), (const OptDev*)e->od, e->upd_snap_small);
            e->fu_parts = true;
        } else
        hipLaunchKernelGGL(k_grad_reduce, dim3(cdiv(e->P, 64)), dim3(1024), 0, e->stream, e->gslab, nred, e->Ppad,
                           e->P, e->grad, add, close ? (const double*)e->dsc() : (const double*)nullptr, assign ? 1 : 0);
    } else {
        const NetDesc& nd = e->nd;
        float* Zb = b.Ob;
        float* nxt = b.A0;
        const float* skip_adj = nullptr;          // zbar of a residual block's second map: joins the adjoint of the block input
        int skip_to = -1;
        for (int lin = nd.n_lin - 1; lin >= 0; --lin) {
            const int K = nd.width[lin], Ho = nd.width[lin + 1];
            const float* Sprev = lin > 0 ? b.S[lin - 1] : nullptr;
            if (nd.skip[lin] >= 0) { skip_adj = Zb; skip_to = nd.skip[lin]; }
            if (lin > 0 && Ho % 128 == 0 && K % 128 == 0 && e->gen_mfma && e->gen_mfma2) {
                // wide hidden->hidden map, 128 x 128 block tiles with LDS-shared operand panels
                const int64_t want = (int64_t)e->num_cu * 4 / ((Ho / 128) * (K / 128)) + 1;       // ~4 blocks per CU
                int64_t chunk = ((b.n + want - 1) / want + 15) / 16 * 16;
                if (chunk < 256) chunk = 256;
                const int64_t nchunk = (b.n + chunk - 1) / chunk;
                dim3 gw(Ho / 128, K / 128, (unsigned)nchunk);
                const size_t pl = (size_t)2 * b.C * 128 * 16 * sizeof(float);
                DISPATCH_TRAIN(b, hipLaunchKernelGGL((g_bwd_weight_mfma2<CC, EE>), gw, dim3(256), pl, e->stream, nd, lin, Sprev, Zb,
                                                    e->grad, b.n, b.ld, chunk));
            } else if (lin > 0 && Ho % 64 == 0 && K % 64 == 0 && e->gen_mfma) {
                // wide hidden->hidden map: split-K GEMM over the points on the matrix cores
                const int64_t want = (int64_t)e->num_cu * 16 / ((Ho / 64) * (K / 64)) + 1;          // chunks so that ~16 waves per CU exist
                int64_t chunk = ((b.n + want - 1) / want + 15) / 16 * 16;
                // (a 64 x 64 map over 4 000 points -- the residual network of refine/box_to_gaussian at its own size -- ran as 16 waves of
                // 256 points each, 100 us per map; measured per step at 256 / 128 / 64 / 32 points per wave: 637 / 447 / 348 / 312 us -- the atomics of a
                // 64 x 64 block cost less than the serial walk)
                if (chunk < e->gen_min_chunk) chunk = e->gen_min_chunk;
                const int64_t nchunk = (b.n + chunk - 1) / chunk;
                dim3 gw(Ho / 64, K / 64, (unsigned)((nchunk + 3) / 4));
                DISPATCH_TRAIN(b, hipLaunchKernelGGL((g_bwd_weight_mfma<CC, EE>), gw, dim3(256), 0, e->stream, nd, lin, Sprev, Zb,
                                                    e->grad, b.n, b.ld, chunk));
            } else if (Ho >= 64 && K >= 16) {   // wide layer: 8 rows of the weight gradient per block share the recomputed jets
                dim3 gw(cdiv(Ho, 8), cdiv(K, G_KB));
                DISPATCH_TRAIN(b, hipLaunchKernelGGL((g_bwd_weight<CC, EE, 8, G_KB>), gw, dim3(256), 0, e->stream, nd, lin, b.pts, Sprev,
                                                    Zb, e->grad, b.n, b.ld));
            } else if (Ho < 8 && K >= 64) {     // output layer of a wide network: one block per (n, k) keeps K blocks in flight
                dim3 gw(Ho, K);
                DISPATCH_TRAIN(b, hipLaunchKernelGGL((g_bwd_weight<CC, EE, 1, 1>), gw, dim3(256), 0, e->stream, nd, lin, b.pts, Sprev,
                                                    Zb, e->grad, b.n, b.ld));
            } else {
                dim3 gw(Ho, cdiv(K, G_KB));
                DISPATCH_TRAIN(b, hipLaunchKernelGGL((g_bwd_weight<CC, EE, 1, G_KB>), gw, dim3(256), 0, e->stream, nd, lin, b.pts, Sprev,
                                                    Zb, e->grad, b.n, b.ld));
            }
            if (e->dp_bucket && close) {
                // data-parallel step: this map's gradient slice (weights + bias, contiguous) is final -> all-reduce it on the
                // exchange stream behind the remaining reverse pass (output map first)
                int rcb = dp_allreduce_after(e, e->grad + nd.offW[lin], (size_t)K * Ho + Ho, ncclFloat, e->ev_bucket[lin]);
                if (rcb) return rcb;
            }
            if (lin > 0) {
                bool act_done = false;
                const bool join = skip_adj && skip_to == lin - 1;      // the skipped-from layer: its adjoint gets the saved zbar first
                if (!join && K % 256 == 0 && Ho % 64 == 0 && e->gen_mfma && e->gen_mfma2) {
                    dim3 gd(cdiv(b.n, 16), K / 256);             // activation adjoint of layer lin-1 fused into the epilogue
                    DISPATCH_TRAIN(b, hipLaunchKernelGGL((g_bwd_data_mfma2<CC, EE>), gd, dim3(256), 0, e->stream, nd, lin, e->theta, Zb,
                                                        nxt, Sprev, b.n, b.ld));
                    act_done = true;
                } else if (K % 64 == 0 && Ho % 16 == 0 && e->gen_mfma) {
                    dim3 gd(cdiv(cdiv(b.n, 16), 4), K / 64);
                    DISPATCH_TRAIN(b, hipLaunchKernelGGL((g_bwd_data_mfma<CC>), gd, dim3(256), 0, e->stream, nd, lin, e->theta, Zb,
                                                        nxt, b.n, b.ld));
                } else if (K >= 64) {
                    dim3 gd(cdiv(b.n, 256), cdiv(K, G_FBW));
                    DISPATCH_TRAIN(b, hipLaunchKernelGGL((g_bwd_data<CC, G_FBW>), gd, dim3(256), 0, e->stream, nd, lin, e->theta, Zb,
                                                        nxt, b.n, b.ld));
                } else {
                    dim3 gd(cdiv(b.n, 256), cdiv(K, G_FB));
                    DISPATCH_TRAIN(b, hipLaunchKernelGGL((g_bwd_data<CC, G_FB>), gd, dim3(256), 0, e->stream, nd, lin, e->theta, Zb,
                                                        nxt, b.n, b.ld));
                }
                if (join) {
                    const int64_t cnt = (int64_t)b.C * K * b.ld;
                    hipLaunchKernelGGL(g_add, dim3(cdiv(cnt, 256)), dim3(256), 0, e->stream, nxt, skip_adj, cnt);
                    skip_adj = nullptr; skip_to = -1;
                }
                if (!act_done) {
                    dim3 ga(cdiv(b.n, 256), K);
                    DISPATCH_TRAIN(b, hipLaunchKernelGGL((g_bwd_act<CC, EE>), ga, dim3(256), 0, e->stream, K, Sprev, nxt, b.n, b.ld));
                }
                Zb = nxt;
                // next free adjoint buffer: not the one just written, not one still waiting for its skip join
                float* cand[3] = {b.A0, b.A1, b.A2};
                for (float* c2 : cand) if (c2 && c2 != Zb && c2 != skip_adj) { nxt = c2; break; }
            }
        }
        if (close) {
            int rc = bc_join(e);
            if (rc) return rc;
            if ((rc = launch_tail(e, true))) return rc;
        }
    }
    HIPCHK(e, hipGetLastError());
    return GPE_OK;
}

// ---- config -> device structs ------------------------------------------------------------------------
static void fill_phys(gpe_engine* e) {
    const gpe_config& c = e->cfg;
    Phys& p = e->ph;
    memset(&p, 0, sizeof p);
    p.dim = c.layers[0]; p.n_out = c.layers[c.n_layers - 1]; p.complex_psi = c.complex_psi;
    p.kin = c.kinetic_coeff; p.potential = c.potential; p.pot_scale = c.pot_scale;
    for (int k = 0; k < 3; ++k) p.omega[k] = c.omega[k];
    p.pot_a = c.pot_a; p.pot_v0 = c.pot_v0; p.pot_k = c.pot_k; p.omega_rot = c.omega_rot;
    p.gamma = c.gamma; p.p = c.p; p.abs_power = c.abs_power; p.base_mode = c.base_mode; p.base_deriv = c.base_deriv;
    p.base_kind = c.base_kind; p.envelope = c.envelope; p.box_L = c.box_L > 0.f ? c.box_L : 1.f; p.env_L = c.env_L > 0.f ? c.env_L : 1.f;
    p.perturb_scale = c.perturb_scale; p.bc_nn_scale = c.bc_nn_scale;
    p.w_pde = c.w_pde; p.w_bc = c.w_bc; p.w_norm = c.w_norm; p.w_sym = c.w_sym; p.w_orth = c.w_orth;
    p.sym_sign = c.sym_sign; p.dx = c.dx; p.w_riesz = c.w_riesz; p.riesz_kind = c.riesz_kind;
    p.lambda_kind = c.lambda_kind; p.w_reg_f = c.w_reg_f; p.reg_f_eps = c.reg_f_eps; p.w_reg_lam = c.w_reg_lam; p.reg_lam_eps = c.reg_lam_eps;
    p.n_global = (double)(c.n_global > 0 ? c.n_global : (e->n_pde > 0 ? e->n_pde : 1));
    p.inv_world = 1.0f / (float)(c.world_size > 0 ? c.world_size : 1);
    int no = 0;
    for (int j = 0; j < GPE_MAX_ORTH; ++j) if (e->orth_host[j]) no = j + 1;      // [4..6] are the precomputed base
    p.n_orth = no;
    if (c.base_mode >= 0) {
        double f = 1.0;
        for (int i = 2; i <= c.base_mode; ++i) f *= i;
        e->base_norm = (float)pow(pow(2.0, c.base_mode) * f * sqrt(M_PI), -0.5);
    }
    OptCfg& o = e->oc;
    o.beta1 = c.beta1; o.beta2 = c.beta2; o.eps = c.eps; o.clip_norm = c.clip_norm; o.sched = c.sched;
    o.T_0 = c.T_0; o.T_mult = c.T_mult; o.eta_min = c.eta_min; o.factor = c.factor; o.patience = c.patience;
    o.min_lr = c.min_lr; o.threshold = c.threshold;
    o.stop_tol = c.stop_tol; o.stop_patience = c.stop_patience;
}

static int reset_opt(gpe_engine* e, float lr) {
    OptDev h;
    memset(&h, 0, sizeof h);
    h.lr = lr; h.lr0 = lr; h.best = INFINITY; h.num_bad = 0; h.nonfinite = 0; h.step = 0;
    h.stopped = 0; h.es_count = 0; h.es_best = INFINITY; h.stop_step = 0; h.b1p = 1.0; h.b2p = 1.0;
    HIPCHK(e, hipMemcpyAsync(e->od, &h, sizeof h, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipMemsetAsync(e->am, 0, (size_t)e->P * sizeof(float), e->stream));
    HIPCHK(e, hipMemsetAsync(e->av, 0, (size_t)e->P * sizeof(float), e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}

extern "C" {

int gpe_abi_version(void) { return GPE_ABI_VERSION; }
size_t gpe_sizeof_config(void) { return sizeof(gpe_config); }
size_t gpe_sizeof_scalars(void) { return sizeof(gpe_scalars); }
int64_t gpe_exchange_dbl_count(void) { return S_COUNT + LS_COUNT + 4; }

int gpe_use_external_exchange(gpe_engine* e, void* d_dbl, int64_t n_dbl, void* d_grad, int64_t n_grad) {
    if (!e || !d_dbl || !d_grad) return GPE_ERR_INVALID;
    if (n_dbl < S_COUNT + LS_COUNT + 4 || n_grad < e->P + GT_COUNT) FAIL(e, GPE_ERR_INVALID, "external exchange buffers too small");
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (!e->ext_exchange) { (void)hipFree(e->dbl); (void)hipFree(e->grad); }
    e->dbl = (double*)d_dbl; e->grad = (float*)d_grad; e->ext_exchange = true;
    e->acc_clean = false;                        // caller-owned buffers: the next step starts with k_begin
    return GPE_OK;
}

const char* gpe_last_error(const gpe_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int gpe_active_path(const gpe_engine* e) { return e ? e->path : GPE_ERR_INVALID; }

// names of the two dominant kernels the bound collocation batch runs (what bench.py prints as roofline.kernel, and what the
// variant tests assert): "fwd=<...>;bwd=<...>"
int gpe_active_kernels(gpe_engine* e, char* buf, size_t n) {
    if (!e || !buf || n == 0) return GPE_ERR_INVALID;
    if (e->main.n <= 0) FAIL(e, GPE_ERR_STATE, "active_kernels before bind_points");
    const Batch& b = e->main;
    char f[160], r[160];
    const int maps = e->nd.n_lin - 2;
    if (e->path == GPE_PATH_FUSED && e->wide) {
        if (e->wide_fwd) snprintf(f, sizeof f, "%s<%d,%d,%d,%d>", wide_forward_kernel(e->H), e->H, b.C, b.E, e->nd.n_out);
        else if (fwd_coop(e, b)) snprintf(f, sizeof f, "f_forward_coop<%d,%d,%d,%d,%d>", e->H, b.C, b.E, e->nd.n_out, maps > 5 ? 5 : maps);
        else snprintf(f, sizeof f, "f_forward<%d,%d,%d,%d,l2>", e->H, b.C, b.E, e->nd.n_out);
        if (wide_reverse(e, const_cast<Batch&>(b))) snprintf(r, sizeof r, "%d x w_bwd_map<%d,%d,%d,%d> (output layer fused into the top map)", maps, e->H, b.C, b.E, e->H / 128);
        else snprintf(r, sizeof r, "f_backward_coop<%d,%d,%d,%d,%d>", e->H, b.C, b.E, e->nd.n_out, maps > 5 ? 5 : maps);
    } else if (e->path == GPE_PATH_FUSED) {
        const bool fc = fwd_coop(e, b) && (e->H <= 64 || b.C <= 4);
        if (fc) snprintf(f, sizeof f, "f_forward_coop<%d,%d,%d,%d,%d>", e->H, b.C, b.E, e->nd.n_out, maps > 5 ? 5 : maps);
        else if (e->H <= 64 && e->fwd_b6 && staged_batch(e, b)) snprintf(f, sizeof f, "f_forward_b6<%d,%d,%d,%d>", e->H, b.C, b.E, e->nd.n_out);
        else snprintf(f, sizeof f, "f_forward<%d,%d,%d,%d,%s>", e->H, b.C, b.E, e->nd.n_out,
                      (e->H <= 64 && e->fwd_wlds && staged_batch(e, b)) ? "wlds" : "l2");
        const int kind = bwd_kind(e, b);
        if (kind == 3 && use_pipe(e, b.C)) snprintf(r, sizeof r, "f_backward_pipe<%d,%d,%d,%d,%d%s>", e->H, b.C, b.E, e->nd.n_out, maps,
                                                    seed_in_reverse(e) ? ",seeds" : "");
        else if (kind == 3) snprintf(r, sizeof r, "f_backward_coop<%d,%d,%d,%d,%d%s>", e->H, b.C, b.E, e->nd.n_out, maps > 5 ? 5 : maps,
                                (e->bwd_b6 && e->H <= 64) ? ",b6" : "");
        else if (kind == 2) snprintf(r, sizeof r, "f_backward<%d,%d,%d,%d,wlds,racc%d>", e->H, b.C, b.E, e->nd.n_out, maps > 3 ? 3 : maps);
        else snprintf(r, sizeof r, "f_backward<%d,%d,%d,%d,l2,%s>", e->H, b.C, b.E, e->nd.n_out, e->H > 64 ? "gacc" : "ldsacc");
        // (whole steps, gpe_step / gpe_run: the split-phase protocol of the data-parallel driver keeps k_head_pde)
        if (fc ? head_fusable_coop(e) : head_fusable_tile(e))
            snprintf(f + strlen(f) - 1, sizeof f - strlen(f) + 1, ",head>");
    } else {
        const int W = e->nd.width[1];
        const bool m2 = e->gen_mfma && e->gen_mfma2 && W % 256 == 0, m1 = e->gen_mfma && W % 64 == 0;
        snprintf(f, sizeof f, "%s<%d,%d>", m2 ? "g_fwd_layer_mfma2" : (m1 ? "g_fwd_layer_mfma" : "g_fwd_layer"), b.C, b.E);
        snprintf(r, sizeof r, "%s<%d,%d>", (e->gen_mfma && e->gen_mfma2 && W % 128 == 0) ? "g_bwd_weight_mfma2"
                                            : (m1 ? "g_bwd_weight_mfma" : "g_bwd_weight"), b.C, b.E);
    }
    // uneven static tile split between the two workgroups of a CU (large batches; launch_f_forward / launch_coop_no apply the same rules)
    int sf = 0, sb = 0;
    if (e->path == GPE_PATH_FUSED && !e->wide && e->H <= 64) {
        const int64_t ntiles = (b.n + 15) / 16, g2 = 2 * (int64_t)e->num_cu;
        if (!fwd_coop(e, b) && !e->fwd_b6 && e->fwd_share > 0 && fused_grid(e, b.n, 4, e->fwd_wg_per_cu) == (unsigned)g2 && ntiles >= 4 * e->share_min_tiles * g2)
            sf = e->fwd_share;
        if (bwd_kind(e, b) == 3 && use_pipe(e, b.C) && !seed_in_reverse(e) && e->pipe_share > 0 &&
            fused_grid(e, b.n, 1, e->coop_wg_per_cu) == (unsigned)g2 && ntiles >= e->share_min_tiles * g2)
            sb = e->pipe_share;
    }
    if (e->umap.empty()) snprintf(buf, n, "fwd=%s;bwd=%s;split=fwd %d/1024, bwd %d/1024", f, r, sf, sb);
    else snprintf(buf, n, "fwd=%s;bwd=%s;split=fwd %d/1024, bwd %d/1024;padded=hidden widths up to %d run as %d", f, r, sf, sb,
                  *std::max_element(e->user_layers + 1, e->user_layers + e->cfg.n_layers - 1), e->cfg.layers[1]);
    return GPE_OK;
}

int64_t gpe_param_count(const gpe_engine* e) { return e ? e->P_user : -1; }   /* the caller's network (hidden widths as given) */

int gpe_create(const gpe_config* cfg, int device, void* hip_stream, gpe_engine** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return GPE_ERR_INVALID; }
    *out = nullptr;
    gpe_engine* e = new gpe_engine();
    auto bail = [&](int code) { g_create_error = e->err; delete e; return code; };
#define CFAIL(...) do { char _b[512]; snprintf(_b, sizeof _b, __VA_ARGS__); e->err = _b; return bail(GPE_ERR_INVALID); } while (0)
    if (cfg->abi_version != GPE_ABI_VERSION) CFAIL("abi_version %d != %d", cfg->abi_version, GPE_ABI_VERSION);
    e->cfg = *cfg;
    const gpe_config& c = e->cfg;
    if (c.n_layers < 3 || c.n_layers > GPE_MAX_LAYERS) CFAIL("n_layers must be in [3,%d]", GPE_MAX_LAYERS);
    const int dim = c.layers[0], no = c.layers[c.n_layers - 1];
    if (dim < 1 || dim > GPE_MAX_DIM) CFAIL("dim must be 1..3");
    if (no < 1 || no > 2) CFAIL("output width must be 1 or 2");
    if (c.complex_psi && (no != 2 || c.p != 3)) CFAIL("complex psi needs out=2 and p=3");
    if (!c.complex_psi && no != 1) CFAIL("real psi needs out=1");
    if (c.potential < 0 || c.potential > GPE_POT_NONE) CFAIL("Unknown potential type: %d", c.potential);
    if (c.base_mode >= 0 && (dim != 1 || no != 1)) CFAIL("Hermite base needs dim=1, out=1");
    if (c.p < 1 || c.p > 32) CFAIL("power p must be in [1,32]");
    if (c.omega_rot != 0.f && (!c.complex_psi || dim < 2)) CFAIL("rotation needs complex psi and dim>=2");
    if (c.base_kind < 0 || c.base_kind > GPE_BASE_PRECOMPUTED) CFAIL("Unknown base kind: %d", c.base_kind);
    if (c.envelope < 0 || c.envelope > GPE_ENV_SIN) CFAIL("Unknown envelope: %d", c.envelope);
    if (c.envelope != GPE_ENV_NONE && (dim != 1 || no != 1)) CFAIL("the boundary factor needs dim=1, out=1");
    if (c.w_riesz != 0.f && no != 1 && !(c.complex_psi && no == 2 && c.p == 3))
        CFAIL("the Riesz energy term needs real psi (out=1) or complex psi (out=2) with p = 3");
    if (c.riesz_kind < 0 || c.riesz_kind > GPE_RIESZ_VARIATIONAL) CFAIL("Unknown Riesz kind: %d", c.riesz_kind);
    if (c.lambda_kind != GPE_LAMBDA_RAYLEIGH && c.lambda_kind != GPE_LAMBDA_ENERGY) CFAIL("Unknown lambda kind: %d", c.lambda_kind);
    if (c.lambda_kind == GPE_LAMBDA_ENERGY && (c.complex_psi || no != 1 || (c.p & 1) == 0))
        CFAIL("the energy-functional lambda needs real psi (out=1) and an odd power p");
    if (c.w_reg_lam != 0.f && c.lambda_kind != GPE_LAMBDA_ENERGY) CFAIL("the 1/lambda^2 regulariser needs the energy-functional lambda");
    if ((c.w_reg_f != 0.f || c.w_reg_lam != 0.f) && (c.complex_psi || no != 1)) CFAIL("the regularisers need real psi (out=1)");
    if ((c.w_reg_f != 0.f && !(c.reg_f_eps > 0.f)) || (c.w_reg_lam != 0.f && !(c.reg_lam_eps > 0.f))) CFAIL("regulariser eps must be > 0");
    for (int i = 1; i < c.n_layers - 1; ++i)
        if (c.layers[i] < 1 || c.layers[i] > 1024) CFAIL("hidden width %d out of range", c.layers[i]);
    // Width padding.  The MFMA kernel sets are instantiated for one hidden width of 32, 64, 128 or 256; other plain-tanh MLPs of width <= 256 --
    // the 2D scripts' own [2,100,100,100,1] (src/gross_pitaevskii_2D_minimal.py:377), ragged widths -- are run as the next such width with
    // zero rows / columns / biases in the padding.  A padded unit sees z = 0, gives tanh(0) = 0 with all jets 0, feeds zero outgoing weights
    // and so receives the gradient 0 exactly on every one of its weights: Adam leaves them at 0 and the network stays the caller's.
    // ShiftedTanh (tanh + 1) would give 1 there: its padded units get the bias -40 instead of 0 -- tanh(-40) = -1 exactly in fp32 (exp(-80) + 1 = 1),
    // so the unit outputs exactly 0 with slope 1 - t^2 = 0, every gradient that touches it is 0 again, and the bias itself stays where it is.
    // GPE_PAD_WIDTH=0 or GPE_PATH_GENERIC: as given.
    for (int i = 0; i < c.n_layers; ++i) e->user_layers[i] = c.layers[i];
    {
        const char* envp = getenv("GPE_PAD_WIDTH");
        int wmax = 0; bool uni = true;
        for (int i = 1; i < c.n_layers - 1; ++i) { wmax = std::max(wmax, c.layers[i]); uni = uni && c.layers[i] == c.layers[1]; }
        const bool native = uni && (wmax == 32 || wmax == 64 || wmax == 128 || wmax == 256);
        // (widths above 256 -- train_pinn's default [2,400,400,400,1], src/gross_pitaevskii_2D_minimal.py:278 -- have no whole-network kernel: they are
        // padded to the next multiple of 256 so that the GENERIC set runs every hidden map on its 128 x 128-tile MFMA kernels instead of the VALU ones)
        const bool big = wmax > 256 && wmax <= 1024 && !(uni && wmax % 256 == 0);
        if (!(envp && atoi(envp) == 0) && (big || (!native && wmax <= 256)) && c.path != GPE_PATH_GENERIC && c.net_kind == GPE_NET_MLP &&
            (c.activation == GPE_ACT_TANH || c.activation == GPE_ACT_TANH_PLUS1) && c.n_layers - 2 >= 2) {
            // the smallest instantiated width that holds the widest layer and whose kernels take this depth (32 / 64: all weights in LDS)
            int Hp = big ? (int)round_up(wmax, 256) : 0;
            if (!big)
            for (int h : {32, 64, 128, 256}) {
                if (h < wmax) continue;
                if (h <= 64) {
                    const int Lm = c.n_layers - 3;                        // hidden -> hidden maps
                    const size_t pp = round_up((size_t)dim * h + h + (size_t)Lm * (h * h + h) + (size_t)h * no + no, 64);
                    if ((pp + (size_t)(8 + c.n_layers + 2) * h + 8 + 4 * (dim + 2) * F_TILE) * sizeof(float) > 160 * 1024) continue;
                }
                Hp = h;
                break;
            }
            int pl[GPE_MAX_LAYERS];
            for (int i = 0; i < c.n_layers; ++i) pl[i] = (i == 0 || i == c.n_layers - 1) ? c.layers[i] : Hp;
            int offp = 0;
            for (int j = 0; j + 1 < c.n_layers; ++j) {
                const int iu = c.layers[j], ou = c.layers[j + 1], ip = pl[j], op = pl[j + 1];
                for (int o = 0; o < ou; ++o) for (int i = 0; i < iu; ++i) e->umap.push_back(offp + o * ip + i);
                offp += ip * op;
                for (int o = 0; o < ou; ++o) e->umap.push_back(offp + o);
                if (c.activation == GPE_ACT_TANH_PLUS1 && j + 2 < c.n_layers) for (int o = ou; o < op; ++o) e->pad_bias.push_back(offp + o);
                offp += op;
            }
            e->P_user = (int)e->umap.size();
            for (int i = 1; i < c.n_layers - 1; ++i) e->cfg.layers[i] = Hp;
        }
    }
    NetDesc& nd = e->nd;
    memset(&nd, 0, sizeof nd);
    nd.dim = dim; nd.n_out = no; nd.shift = c.activation == GPE_ACT_TANH_PLUS1 ? 1.f : 0.f;
    if (c.net_kind != GPE_NET_MLP && c.net_kind != GPE_NET_RESIDUAL) CFAIL("Unknown network kind: %d", c.net_kind);
    for (int j = 0; j < GPE_MAX_LAYERS; ++j) { nd.skip[j] = -1; nd.shiftv[j] = nd.shift; }
    if (c.net_kind == GPE_NET_RESIDUAL) {      // refine/box_to_gaussian_pinn_simulation.py:100-130
        const int nb = c.n_layers - 3;
        if (nb < 1 || 2 * nb + 3 > GPE_MAX_LAYERS) CFAIL("residual network: 1..%d blocks", (GPE_MAX_LAYERS - 3) / 2);
        for (int i = 2; i < c.n_layers - 1; ++i) if (c.layers[i] != c.layers[1]) CFAIL("residual network: one hidden width");
        nd.n_lin = 2 * nb + 2;
        nd.width[0] = dim;
        for (int i = 1; i <= 2 * nb + 1; ++i) nd.width[i] = c.layers[1];
        nd.width[nd.n_lin] = no;
        for (int bq = 0; bq < nb; ++bq) nd.skip[2 * bq + 2] = 2 * bq;     // lin2 of block bq adds hidden layer 2 bq (the block input)
        for (int h = 1; h < nd.n_lin - 1; ++h) nd.shiftv[h] = 0.f;        // plain tanh inside the blocks; the first layer keeps `activation`
    } else {
        nd.n_lin = c.n_layers - 1;
        for (int i = 0; i < c.n_layers; ++i) nd.width[i] = c.layers[i];
    }
    int off = 0;
    for (int j = 0; j < nd.n_lin; ++j) {
        nd.offW[j] = off; off += nd.width[j] * nd.width[j + 1];
        nd.offB[j] = off; off += nd.width[j + 1];
    }
    nd.n_params = off;
    e->P = off;
    if (e->umap.empty()) e->P_user = off;
    e->Ppad = (int)round_up(off, 64);
    // path selection: fused kernels need >= 2 hidden layers of one width H in {32, 64}
    bool uniform = true;
    for (int i = 2; i < c.n_layers - 1; ++i) uniform = uniform && (c.layers[i] == c.layers[1]);
    const int H = c.layers[1];
    const int Lh = nd.n_lin - 1;                                                 // hidden layers (an MLP's n_layers - 2; a residual network's 2 blocks + 1)
    size_t lds_need = ((size_t)e->Ppad + (size_t)(8 + nd.n_lin + 3) * c.layers[1] + 8 + 4 * (dim + 2) * F_TILE) * sizeof(float);
    // residual blocks (refine/box_to_gaussian_pinn_simulation.py): round 4 -- one or two blocks of width 32 / 64, real psi, on the cooperative
    // whole-network kernels (f_forward_coop / f_backward_coop <..., RES>); everything else on the generic set
    const bool residual = c.net_kind == GPE_NET_RESIDUAL;
    if (residual && !((H == 32 || H == 64) && no == 1 && (nd.n_lin - 2 == 2 || nd.n_lin - 2 == 4))) uniform = false;
    { const char* envr = getenv("GPE_RES_FUSED"); if (residual && envr && atoi(envr) == 0) uniform = false; }
    bool fused_ok = uniform && (H == 32 || H == 64) && Lh >= 2 && lds_need <= 160 * 1024;
    if (residual && (H == 128 || H == 256)) uniform = false;                    // (no residual form of the 8-wave kernels)
    if (uniform && H == 128 && Lh >= 2 && dim <= 2) fused_ok = true;          // cooperative kernels, weights streamed from L2
    // wide kernel set: H = 256 and 3D H = 128 entirely; 1D/2D H = 128: its per-map reverse kernels (no register spills, -5..7 % against
    // f_backward_coop<128>) behind the cooperative forward kernel (7 % faster than w_forward there) -- same stored-activation format.
    // GPE_WIDE=1: forward too; GPE_WIDE=0: cooperative kernels only (1D/2D)
    const char* envw = getenv("GPE_WIDE");
    const int wmode = envw ? atoi(envw) : -1;
    if (uniform && Lh >= 2 && (H == 256 || (H == 128 && dim == 3))) { fused_ok = true; e->wide = true; e->wide_fwd = true; }
    else if (uniform && Lh >= 2 && H == 128 && wmode != 0) { fused_ok = true; e->wide = true; e->wide_fwd = wmode == 1; }
    if (c.path == GPE_PATH_FUSED && !fused_ok)
        CFAIL("fused path needs >=2 hidden layers of one width: 32 or 64 (P*4 <= 160KB LDS), 128 or 256");
    e->path = (c.path == GPE_PATH_GENERIC || !fused_ok) ? GPE_PATH_GENERIC : GPE_PATH_FUSED;
    e->H = H;
#undef CFAIL
    e->device = device;
    e->stream = (hipStream_t)hip_stream;
    hipError_t st = hipSetDevice(device);
    if (st != hipSuccess) { e->err = std::string("hipSetDevice: ") + hipGetErrorString(st); return bail(GPE_ERR_HIP); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) e->num_cu = prop.multiProcessorCount;
    e->cap = c.history_capacity > 0 ? c.history_capacity : 65536;
    auto alloc = [&](void** p, size_t bytes) -> bool {
        hipError_t s2 = hipMalloc(p, bytes + 256);
        if (s2 != hipSuccess) { e->err = std::string("hipMalloc: ") + hipGetErrorString(s2); return false; }
        (void)hipMemset(*p, 0, bytes + 256);
        return true;
    };
    bool ok = alloc((void**)&e->theta, (size_t)e->P * 4) && alloc((void**)&e->am, (size_t)e->P * 4) &&
              alloc((void**)&e->av, (size_t)e->P * 4) && alloc((void**)&e->grad, ((size_t)e->P + GT_COUNT) * 4) &&
              alloc((void**)&e->dbl, (S_COUNT + LS_COUNT + 4) * sizeof(double)) && alloc((void**)&e->od, sizeof(OptDev)) &&
              alloc((void**)&e->hist, (size_t)e->cap * sizeof(gpe_scalars)) && alloc((void**)&e->last, sizeof(gpe_scalars)) &&
              alloc((void**)&e->orth_dev, 8 * sizeof(float*));
    {
        const char* envu = getenv("GPE_UPDATE_MULTI");               // 0: the single-workgroup update at every size
        const char* envum = getenv("GPE_UPDATE_MULTI_MIN");          // tests: the multi-workgroup form from this many parameters on
        const int multi_min = envum ? atoi(envum) : UPD_MULTI_MIN;
        if (ok && e->P >= multi_min && !(envu && atoi(envu) == 0)) ok = alloc((void**)&e->upd_snap, sizeof(UpdSnap));
        const char* envuc = getenv("GPE_UPDATE_CACHE");
        e->update_cache = !(envuc && atoi(envuc) == 0);
        const char* envf3 = getenv("GPE_FUSE_SEED");
        e->fuse_seed = !(envf3 && atoi(envf3) == 0);
        const char* envf4 = getenv("GPE_FUSE_SEED_MAX");
        if (envf4) e->fuse_seed_max = atoll(envf4);
        const char* envps = getenv("GPE_PIPE_SHARE");
        if (envps) e->pipe_share = atoi(envps);
        const char* envsm = getenv("GPE_SHARE_MIN_TILES");
        if (envsm && atoi(envsm) > 0) e->share_min_tiles = atoi(envsm);
        const char* envfs = getenv("GPE_FWD_SHARE");
        if (envfs) e->fwd_share = atoi(envfs);
        const char* envf5 = getenv("GPE_FUSE_HEAD");
        e->fuse_head = !(envf5 && atoi(envf5) == 0);
        const char* envf7 = getenv("GPE_FUSE_HEAD_TILE_MIN");
        if (envf7) e->fuse_head_tile_min = atoll(envf7);
        const char* envf6 = getenv("GPE_FUSE_HEAD_MAX");
        if (envf6) e->fuse_head_max = atoll(envf6);
        if (ok && e->fuse_head) ok = alloc((void**)&e->head_slots, (size_t)HEAD_SLOTS * 4 * sizeof(double));
        // opt-in: measured 1 us SLOWER than the two launches at 2 048 .. 65 536 points (profiles/r04/small_batch.txt) -- 200 workgroups'
        // tickets on one address (~12 ns apiece) and the write-through hand-off cost what the saved launch gives
        const char* envfu = getenv("GPE_FUSE_UPDATE");
        e->fuse_update = envfu && atoi(envfu) != 0;
        if (ok && e->fuse_update) ok = alloc((void**)&e->upd_ticket, 64);
        // opt-in as well: measured EQUAL to the single-workgroup update (35.8 vs 35.7 us at 2 048 points, 41.4 vs 41.0 at 4 000,
        // profiles/r04/small_batch.txt) -- at these sizes a dependent launch costs ~4 us whatever it runs, and the update's own chain is short
        const char* envsu = getenv("GPE_SPLIT_UPDATE");
        e->split_update = envsu && atoi(envsu) != 0;
        if (ok && e->split_update && !e->upd_snap && upd_chunk_host(e->P) <= 1024) ok = alloc((void**)&e->upd_snap_small, sizeof(UpdSnap));

    }
    if (ok && e->path == GPE_PATH_FUSED) {
        e->nslab = (H >= 128) ? std::max(e->nslab_g, e->num_cu) : e->num_cu * 2;     // H = 128: 16 atomic slabs, or one per workgroup (cooperative)
        if (e->wide) wide_init();
        { const char* envm = getenv("GPE_WIDE_MIN_TILES"); if (envm && atoll(envm) >= 0) e->wide_min_tiles = atoll(envm); }
        // WpkT is followed by the bf16 pieces of the same maps (3 x 2 bytes per weight; pack_weight_element, f_forward_b6)
        ok = alloc((void**)&e->Wpk, (size_t)(Lh - 1) * H * H * 4) && alloc((void**)&e->WpkT, (size_t)(Lh - 1) * H * H * (4 + 6 + 6)) &&
             alloc((void**)&e->gslab, (size_t)e->nslab * e->Ppad * 4);
        if (ok) {
            const int Cmain = dim + 2;        // training batches: value, dim first derivatives, Laplacian
            const size_t wb = (size_t)(Lh - 1) * H * H * sizeof(float);
            const char* env = getenv("GPE_WLDS");                     // tuning switch: 0 = weights from L2, 1 = from LDS
            const bool want = env ? (atoi(env) != 0) : true;
            const size_t smallb = ((size_t)(4 + (Lh - 1) + no) * H + 8) * sizeof(float);
            e->fwd_wlds = want && H <= 64 && wb + smallb <= 64 * 1024;
            const char* envb6 = getenv("GPE_FWD_B6");                  // 1: f_forward_b6 for the large-batch forward pass at H <= 64
            e->fwd_b6 = envb6 && atoi(envb6) != 0 && H <= 64;
            const char* envb7 = getenv("GPE_BWD_B6");                  // 1: the cooperative reverse kernel's adjoint products on the bf16 pipe
            e->bwd_b6 = envb7 && atoi(envb7) != 0 && H <= 64 && H >= 32;
            const char* envp = getenv("GPE_PIPE");                     // 0: the two-barrier cooperative reverse kernel (f_backward_coop) at H <= 64
            e->bwd_pipe = !envp || atoi(envp) != 0;
            const char* envw = getenv("GPE_COOP_WG_PER_CU");
            if (envw && atoi(envw) >= 1 && atoi(envw) <= 2) e->coop_wg_per_cu = atoi(envw);
            const char* envfw = getenv("GPE_FWD_WG_PER_CU");
            if (envfw && atoi(envfw) >= 1 && atoi(envfw) <= 4) e->fwd_wg_per_cu = atoi(envfw);
            const char* envc = getenv("GPE_COOP");
            e->coop = envc ? atoi(envc) : 1;          // measured: faster than the per-wave-tile kernels at every batch size
            const char* envm = getenv("GPE_COOP_MAX_TILES");
            e->coop_max_tiles = envm ? atoll(envm) : 0;
            const char* env8 = getenv("GPE_COOP128");
            e->coop128 = !env8 || atoi(env8) != 0;
            const char* env9 = getenv("GPE_COOP_FWD128");
            e->coop_fwd128 = !env9 || atoi(env9) != 0;
            const char* envf = getenv("GPE_COOP_FWD_MAX_TILES");
            e->coop_fwd_max_tiles = envf ? atoll(envf) : (int64_t)e->num_cu * 8;   // measured: wins below ~32 768 points, loses 8 % at 1M
            const char* envs = getenv("GPE_STAGE_MIN_TILES");        // batches with fewer tiles take the unstaged per-wave-tile kernels
            e->stage_min_tiles = envs ? atoll(envs) : 0;
            const char* envr = getenv("GPE_RACC");
            e->bwd_racc = (!envr || atoi(envr) != 0) && H <= 64 && (Lh - 1) >= 1 && (Lh - 1) <= 3 &&
                          ((size_t)e->Ppad + 4 * (size_t)H + 4 * (size_t)Cmain * F_TILE) * sizeof(float) + smallb + wb <= 160 * 1024;
            // allow > 64 KB dynamic LDS
            const int lds_b = 160 * 1024, lds_f = 64 * 1024;
#define SETLDS(HH, CC, EE, NO)                                                                                                   \
    (void)hipFuncSetAttribute((const void*)f_backward<HH, CC, EE, NO, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_backward<HH, CC, EE, NO, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);  \
    (void)hipFuncSetAttribute((const void*)f_backward<HH, CC, EE, NO, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);  \
    (void)hipFuncSetAttribute((const void*)f_backward<HH, CC, EE, NO, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);  \
    (void)hipFuncSetAttribute((const void*)f_backward_coop<HH, CC, EE, NO, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_backward_coop<HH, CC, EE, NO, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_backward_coop<HH, CC, EE, NO, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    if constexpr (NO == 1) {                                                                                                        \
        (void)hipFuncSetAttribute((const void*)f_backward_coop<HH, CC, EE, 1, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
        (void)hipFuncSetAttribute((const void*)f_backward_coop<HH, CC, EE, 1, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
        (void)hipFuncSetAttribute((const void*)f_backward_coop<HH, CC, EE, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
        (void)hipFuncSetAttribute((const void*)f_backward_coop<HH, CC, EE, 1, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    }                                                                                                                               \
    set_pipe_lds<HH, CC, EE, NO>(lds_b);                                                                                        \
    (void)hipFuncSetAttribute((const void*)f_backward_coop<HH, CC, EE, NO, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_backward_coop<HH, CC, EE, NO, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_backward_coop<HH, CC, EE, NO, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_forward<HH, CC, EE, NO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_f);      \
    (void)hipFuncSetAttribute((const void*)f_forward_b6<HH, CC, EE, NO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)
#ifdef GPE_FAST_BUILD
            SETLDS(64, 1, 0, 1); SETLDS(64, 4, 1, 1);
#else
            SETLDS(64, 1, 0, 1); SETLDS(64, 3, 1, 1); SETLDS(64, 4, 1, 1); SETLDS(64, 5, 1, 1);
            SETLDS(64, 1, 0, 2); SETLDS(64, 3, 1, 2); SETLDS(64, 4, 1, 2); SETLDS(64, 5, 1, 2);
            SETLDS(32, 1, 0, 1); SETLDS(32, 3, 1, 1); SETLDS(32, 4, 1, 1); SETLDS(32, 5, 1, 1);
            SETLDS(32, 1, 0, 2); SETLDS(32, 3, 1, 2); SETLDS(32, 4, 1, 2); SETLDS(32, 5, 1, 2);
#undef SETLDS
#define SETLDS128(CC, EE, NO)                                                                                                  \
    (void)hipFuncSetAttribute((const void*)f_backward_coop<128, CC, EE, NO, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_backward_coop<128, CC, EE, NO, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_backward_coop<128, CC, EE, NO, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_backward_coop<128, CC, EE, NO, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_backward_coop<128, CC, EE, NO, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_forward_coop<128, CC, EE, NO, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_forward_coop<128, CC, EE, NO, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_forward_coop<128, CC, EE, NO, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_forward_coop<128, CC, EE, NO, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_forward_coop<128, CC, EE, NO, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); \
    (void)hipFuncSetAttribute((const void*)f_backward<128, CC, EE, NO, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b)
            SETLDS128(1, 0, 1); SETLDS128(3, 1, 1); SETLDS128(4, 1, 1); SETLDS128(1, 0, 2); SETLDS128(3, 1, 2); SETLDS128(4, 1, 2);
#undef SETLDS128
#endif
        }
    }
    if (ok) {
        ok = alloc((void**)&e->grad_bc, (size_t)e->P * 4);
        if (ok && e->path == GPE_PATH_FUSED) ok = alloc((void**)&e->gslab_bc, (size_t)e->nslab * e->Ppad * 4);
        const char* envm2 = getenv("GPE_GEN_MFMA");
        e->gen_mfma = !envm2 || atoi(envm2) != 0;
        { const char* envc = getenv("GPE_GEN_MIN_CHUNK"); if (envc && atoll(envc) >= 16) e->gen_min_chunk = (atoll(envc) + 15) / 16 * 16; }
        const char* envm3 = getenv("GPE_GEN_MFMA2");
        e->gen_mfma2 = !envm3 || atoi(envm3) != 0;
#ifndef GPE_FAST_BUILD
        (void)hipFuncSetAttribute((const void*)g_bwd_weight_mfma2<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)g_bwd_weight_mfma2<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)g_bwd_weight_mfma2<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)g_bwd_weight_mfma2<5, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#else
        (void)hipFuncSetAttribute((const void*)g_bwd_weight_mfma2<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)g_bwd_weight_mfma2<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
        const char* envh = getenv("GPE_HEAD_WG_PER_CU");
        if (envh && atoi(envh) > 0) e->head_wg_per_cu = atoi(envh);
        const char* envht = getenv("GPE_HEAD_THREADS");
        if (envht && (atoi(envht) == 256 || atoi(envht) == 512 || atoi(envht) == 1024)) e->head_threads = atoi(envht);
        const char* envb = getenv("GPE_MERGE_BC");
        e->merge_bc = !envb || atoi(envb) != 0;
        const char* envg = getenv("GPE_GRAPH");
        // gpe_run replays captured graphs of graph_steps steps each.  Round 4: ONE graph launch per 8 steps takes the host from 31 to
        // 3 us per step at 2 048 points with the step time unchanged (35.5 us: the GPU is the bound); a one-step graph is slower (44.8 us).
        // Default: on for batches up to graph_max_points (where the host thread was ~90 % busy enqueueing); GPE_GRAPH=0 / 1 forces it
        e->use_graph = envg ? atoi(envg) != 0 : true;
        e->graph_forced = envg != nullptr;
        const char* envgs = getenv("GPE_GRAPH_STEPS");
        if (envgs && atoi(envgs) >= 1 && atoi(envgs) <= 64) e->graph_steps = atoi(envgs);
        const char* envq = getenv("GPE_SIDE_STREAM");
        if (ok && (!envq || atoi(envq) != 0)) {
            if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess) {
                e->err = "side stream / events could not be created";
                return bail(GPE_ERR_HIP);
            }
        }
    }
    if (!ok) return bail(GPE_ERR_NOMEM);
    fill_phys(e);
    int rc = reset_opt(e, c.lr);
    if (rc) return bail(rc);
    if (!e->pad_bias.empty()) {                   // (the padding's biases are in place before the first gpe_set_params, too)
        std::vector<float> pb(e->pad_bias.size(), -40.f);
        for (size_t i = 0; i < pb.size(); ++i)
            if (hipMemcpy(e->theta + e->pad_bias[i], &pb[i], sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { e->err = "hipMemcpy (padding)"; return bail(GPE_ERR_HIP); }
    }
    *out = e;
    return GPE_OK;
}

static void graph_drop(gpe_engine* e);
int gpe_comm_destroy(gpe_engine* e);

void gpe_destroy(gpe_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    if (e->side) { (void)hipStreamSynchronize(e->side); (void)hipStreamDestroy(e->side); }
    (void)gpe_comm_destroy(e);
    graph_drop(e);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    free_batch(e->main); free_batch(e->bc); free_batch(e->sym); free_batch(e->aux); free_batch(e->mse);
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->ext_exchange) { e->grad = nullptr; e->dbl = nullptr; }
    void* ps[] = {e->theta, e->am, e->av, e->grad, e->dbl, e->od, e->hist, e->last, (void*)e->orth_dev, e->Wpk, e->WpkT, e->gslab, e->gslab_bc, e->grad_bc, (void*)e->upd_snap, (void*)e->head_slots, (void*)e->upd_ticket, (void*)e->upd_snap_small};
    for (void* p : ps) if (p) (void)hipFree(p);
    delete e;
}

int gpe_synchronize(gpe_engine* e) {
    if (!e) return GPE_ERR_INVALID;
    if (e->side) HIPCHK(e, hipStreamSynchronize(e->side));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}

// caller's flat vector (P_user entries, hidden widths as given) <-> the engine's (P entries, padded widths)
static int flat_to_device(gpe_engine* e, float* d_dst, const float* h_user) {
    if (e->umap.empty()) { HIPCHK(e, hipMemcpyAsync(d_dst, h_user, (size_t)e->P * 4, hipMemcpyHostToDevice, e->stream)); return GPE_OK; }
    std::vector<float> tmp((size_t)e->P, 0.f);
    if (d_dst == e->theta) for (int i : e->pad_bias) tmp[i] = -40.f;         // (parameters only: the Adam moments of the padding are 0)
    for (int i = 0; i < e->P_user; ++i) tmp[e->umap[i]] = h_user[i];
    HIPCHK(e, hipMemcpyAsync(d_dst, tmp.data(), (size_t)e->P * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));              // tmp dies with this frame
    return GPE_OK;
}
static int flat_from_device(gpe_engine* e, const float* d_src, float* h_user) {
    if (e->umap.empty()) { HIPCHK(e, hipMemcpyAsync(h_user, d_src, (size_t)e->P * 4, hipMemcpyDeviceToHost, e->stream)); return GPE_OK; }
    std::vector<float> tmp((size_t)e->P);
    HIPCHK(e, hipMemcpyAsync(tmp.data(), d_src, (size_t)e->P * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (int i = 0; i < e->P_user; ++i) h_user[i] = tmp[e->umap[i]];
    return GPE_OK;
}

int gpe_set_params(gpe_engine* e, const float* h, size_t n) {
    if (!e || !h) return GPE_ERR_INVALID;
    if ((int64_t)n != e->P_user) FAIL(e, GPE_ERR_INVALID, "set_params: got %zu floats, model has %d", n, e->P_user);
    int rc = flat_to_device(e, e->theta, h);
    if (rc) return rc;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->packed_dirty = true;
    return GPE_OK;
}
int gpe_get_params(gpe_engine* e, float* h, size_t n) {
    if (!e || !h) return GPE_ERR_INVALID;
    if ((int64_t)n != e->P_user) FAIL(e, GPE_ERR_INVALID, "get_params: got %zu floats, model has %d", n, e->P_user);
    int rc = flat_from_device(e, e->theta, h);
    if (rc) return rc;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}
int gpe_get_grad(gpe_engine* e, float* h, size_t n) {
    if (!e || !h) return GPE_ERR_INVALID;
    if ((int64_t)n != e->P_user) FAIL(e, GPE_ERR_INVALID, "get_grad: size mismatch");
    int rc = flat_from_device(e, e->grad, h);
    if (rc) return rc;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}
int gpe_get_adam_state(gpe_engine* e, float* hm, float* hv, size_t n, int64_t* step) {
    if (!e) return GPE_ERR_INVALID;
    if ((int64_t)n != e->P_user) FAIL(e, GPE_ERR_INVALID, "get_adam_state: size mismatch");
    OptDev h;
    int rc;
    if (hm && (rc = flat_from_device(e, e->am, hm))) return rc;
    if (hv && (rc = flat_from_device(e, e->av, hv))) return rc;
    HIPCHK(e, hipMemcpyAsync(&h, e->od, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (step) *step = h.step;
    return GPE_OK;
}
int gpe_set_adam_state(gpe_engine* e, const float* hm, const float* hv, size_t n, int64_t step) {
    if (!e || !hm || !hv) return GPE_ERR_INVALID;
    if ((int64_t)n != e->P_user) FAIL(e, GPE_ERR_INVALID, "set_adam_state: size mismatch");
    OptDev h;
    HIPCHK(e, hipMemcpyAsync(&h, e->od, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    h.step = step;
    h.b1p = pow((double)e->oc.beta1, (double)step); h.b2p = pow((double)e->oc.beta2, (double)step);
    int rc;
    if ((rc = flat_to_device(e, e->am, hm))) return rc;
    if ((rc = flat_to_device(e, e->av, hv))) return rc;
    HIPCHK(e, hipMemcpyAsync(e->od, &h, sizeof h, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}
int gpe_reset_optimizer(gpe_engine* e, float lr) {
    if (!e) return GPE_ERR_INVALID;
    e->cfg.lr = lr;
    return reset_opt(e, lr);
}

// (re)build the collocation batch from what is bound; decides whether the boundary points ride in it
static int rebuild_main(gpe_engine* e) {
    if (!e->ux) return GPE_OK;
    const bool merged = e->merge_bc && e->uxb && e->nb_user > 0 && e->nb_user * 8 <= e->n_pde;
    const int64_t n_tot = e->n_pde + (merged ? e->nb_user : 0);
    int rc = setup_batch(e, e->main, e->ux, n_tot, e->nd.dim + 2, 1, true, e->uV);     // value, dim first derivatives, Laplacian
    if (rc) return rc;
    e->main.pts = Pts{e->ux, merged ? e->uxb : nullptr, e->n_pde};
    e->nb_merged = merged ? e->nb_user : 0;
    if (merged) free_batch(e->bc);
    else if (e->uxb && e->nb_user > 0) { if ((rc = setup_batch(e, e->bc, e->uxb, e->nb_user, 1, 0, false, nullptr))) return rc; }
    else free_batch(e->bc);
    return GPE_OK;
}

int gpe_bind_points(gpe_engine* e, const float* d_x, int64_t n_local, const float* d_V) {
    if (!e) return GPE_ERR_INVALID;
    if (!d_x || n_local <= 0) FAIL(e, GPE_ERR_INVALID, "bind_points: need n_local > 0 points");
    if (e->cfg.potential == GPE_POT_PRECOMPUTED && !d_V) FAIL(e, GPE_ERR_INVALID, "precomputed potential requested but d_V is NULL");
    e->ux = d_x; e->uV = d_V; e->n_pde = n_local;
    int rc = rebuild_main(e);
    if (rc) return rc;
    if (e->cfg.w_sym != 0.f) {
        // symmetry batch: [x ; -x], value only
        if (e->sym.n != 2 * n_local) {
            free_batch(e->sym);
            float* xs = nullptr;
            Batch tmp;
            if ((rc = dev_alloc(e, &tmp, &xs, (size_t)2 * n_local * e->nd.dim))) return rc;
            if ((rc = setup_batch(e, e->sym, xs, 2 * n_local, 1, 0, false, nullptr))) return rc;
            e->sym.allocs.push_back(tmp.allocs[0]);
            e->sym.xown = xs;
        }
        hipLaunchKernelGGL(k_make_sym_points, dim3(cdiv(n_local * e->nd.dim, 256)), dim3(256), 0, e->stream, d_x,
                           e->sym.xown, n_local, e->nd.dim);
        HIPCHK(e, hipGetLastError());
    }
    fill_phys(e);
    e->acc_clean = false;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}

int gpe_bind_boundary(gpe_engine* e, const float* d_xb, int64_t n_b, const float* d_target) {
    if (!e) return GPE_ERR_INVALID;
    if (!d_xb || n_b <= 0) { e->uxb = nullptr; e->nb_user = 0; e->bc_target = nullptr; }
    else { e->uxb = d_xb; e->nb_user = n_b; e->bc_target = d_target; }
    if (e->ux) {
        int rc = rebuild_main(e);
        if (rc) return rc;
    } else if (e->uxb) {             // points not bound yet: keep a plain boundary batch until they are
        int rc = setup_batch(e, e->bc, e->uxb, e->nb_user, 1, 0, false, nullptr);
        if (rc) return rc;
    } else free_batch(e->bc);
    e->acc_clean = false;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}

int gpe_bind_orth(gpe_engine* e, int k, const float* d_psi) {
    if (!e || k < 0 || k >= GPE_MAX_ORTH) return GPE_ERR_INVALID;
    e->orth_host[k] = d_psi;
    HIPCHK(e, hipMemcpyAsync((void*)e->orth_dev, e->orth_host, sizeof e->orth_host, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    fill_phys(e);
    return GPE_OK;
}

int gpe_bind_base(gpe_engine* e, const float* d_phi, const float* d_phi1, const float* d_phi2) {
    if (!e) return GPE_ERR_INVALID;
    e->orth_host[4] = d_phi; e->orth_host[5] = d_phi1; e->orth_host[6] = d_phi2;
    HIPCHK(e, hipMemcpyAsync((void*)e->orth_dev, e->orth_host, sizeof e->orth_host, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}

// ---- forward-only ----------------------------------------------------------------------------------
static int aux_forward(gpe_engine* e, const float* d_x, int64_t n, int C, int E) {
    int rc = setup_batch(e, e->aux, d_x, n, C, E, true, nullptr, /*with_store=*/false);
    if (rc) return rc;
    return mlp_forward(e, e->aux, false);
}

int gpe_forward(gpe_engine* e, const float* d_x, int64_t n, float* d_out) {
    if (!e || !d_x || !d_out || n <= 0) return GPE_ERR_INVALID;
    int rc = aux_forward(e, d_x, n, 1, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(k_copy_values, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, e->ph, d_x, e->aux.O, d_out, n, e->aux.ld, e->nd.n_out);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}

int gpe_forward_jets(gpe_engine* e, const float* d_x, int64_t n, float* d_jets) {
    if (!e || !d_x || !d_jets || n <= 0) return GPE_ERR_INVALID;
    const int C = 1 + 2 * e->nd.dim;
    int rc = aux_forward(e, d_x, n, C, e->nd.dim);        // full diagonal second derivatives: E = dim
    if (rc) return rc;
    hipLaunchKernelGGL(k_copy_jets, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, e->aux.O, d_jets, n, e->aux.ld, e->nd.n_out, C);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}

int gpe_eval_density(gpe_engine* e, const float* d_x, int64_t n, float dx, int abs_flag, float* d_u, float* d_dens) {
    if (!e || !d_x || n <= 0) return GPE_ERR_INVALID;
    if (e->cfg.base_mode >= 0 && e->cfg.base_kind == GPE_BASE_PRECOMPUTED)
        FAIL(e, GPE_ERR_INVALID, "eval_density needs an analytic base (the precomputed base exists on the bound points only)");
    int rc = aux_forward(e, d_x, n, 1, 0);
    if (rc) return rc;
    double* acc = e->dsc() + 1;
    e->acc_clean = false;
    HIPCHK(e, hipMemsetAsync(acc, 0, sizeof(double), e->stream));
    hipLaunchKernelGGL(k_eval_u, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, e->ph, e->base_norm, d_x, e->aux.O, e->aux.u,
                       acc, n, e->aux.ld);
    hipLaunchKernelGGL(k_eval_finish, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, e->aux.u, acc, dx, abs_flag, d_u, d_dens,
                       n, e->aux.ld, e->nd.n_out);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}

// ---- the step ------------------------------------------------------------------------------------------
static unsigned head_grid(gpe_engine* e, int64_t n, int threads = 0) {
    return (unsigned)std::min<int64_t>(cdiv(n, threads > 0 ? threads : e->head_threads), (int64_t)e->num_cu * e->head_wg_per_cu);
}

static int launch_head_pde(gpe_engine* e) {
    Batch& b = e->main;
    dim3 g(head_grid(e, b.n));
    DISPATCH_TRAIN(b, hipLaunchKernelGGL((k_head_pde<CC, EE>), g, dim3(e->head_threads), 0, e->stream, e->ph, e->base_norm, b.pts, b.V, b.O,
                                        (const float* const*)e->orth_dev, b.u, b.Hu, b.ux, e->sums(), b.n, b.ld, e->n_pde,
                                        e->bc_target, b.Ob, e->lsums()));
    HIPCHK(e, hipGetLastError());
    return GPE_OK;
}
static int launch_seed_pde(gpe_engine* e, float* d_resid, int want_seeds) {
    Batch& b = e->main;
    dim3 g(head_grid(e, e->n_pde));
    DISPATCH_TRAIN(b, hipLaunchKernelGGL((k_seed_pde<CC, EE>), g, dim3(e->head_threads), 0, e->stream, e->ph, b.pts, b.V,
                                        (const float* const*)e->orth_dev, b.u, b.Hu, b.ux, e->sums(), b.Ob, d_resid, e->dsc(),
                                        e->n_pde, b.ld, want_seeds, e->fh_now ? (const double*)e->head_slots : (const double*)nullptr,
                                        e->fh_now ? e->fh_nslots : 0, e->sums(), e->lsums()));      // collocation rows only; boundary rows were seeded by the head kernel
    HIPCHK(e, hipGetLastError());
    return GPE_OK;
}

static int bc_fork(gpe_engine* e, bool with_backward);

int gpe_step_begin(gpe_engine* e) {
    if (!e) return GPE_ERR_INVALID;
    if (e->main.n <= 0) FAIL(e, GPE_ERR_STATE, "step before bind_points");
    if (e->cfg.base_mode >= 0 && e->cfg.base_kind == GPE_BASE_PRECOMPUTED && !e->orth_host[4])
        FAIL(e, GPE_ERR_STATE, "precomputed base requested but gpe_bind_base was not called");
    int rc;
    e->fh_now = false; e->seedf_now = false; e->fu_done = false; e->fu_parts = false;      // per-step flags: a step that failed half-way must not leave them behind
    e->phase = 0;
    if ((rc = launch_begin(e))) return rc;
    if ((rc = bc_fork(e, true))) return rc;
    const bool fh = head_in_forward(e);
    if (fh) e->fh_nslots = (int)(head_fusable_coop(e) ? fused_grid(e, e->main.n, 1, 2) : fused_grid(e, e->main.n, 4, e->fwd_wg_per_cu));
    e->fh_now = fh;
    rc = mlp_forward(e, e->main, true);
    if (!rc && !fh) rc = launch_head_pde(e);
    if (!rc && e->cfg.w_sym != 0.f) {
        if (!(rc = mlp_forward(e, e->sym, true))) {
            hipLaunchKernelGGL(k_head_sym, dim3(cdiv(e->n_pde, 256)), dim3(256), 0, e->stream, e->ph, e->sym.O, e->sums(),
                               e->n_pde, e->sym.ld);
            if (hipGetLastError() != hipSuccess) { e->err = "k_head_sym launch failed"; rc = GPE_ERR_HIP; }
        }
    }
    if (rc) { e->fh_now = false; return rc; }
    e->phase = 1;
    return GPE_OK;
}

// Boundary batch (refine/harmonic_pinn_simulation.py:197-210): forward, loss + seeds, reverse.  It does not depend on mu, so it
// is forked onto the side stream at the start of the step and overlaps the collocation batch; its gradient lands in grad_bc
// and k_tail adds it after the join.
static int bc_fork(gpe_engine* e, bool with_backward) {
    e->bc_inflight = false;
    if (e->bc.n <= 0 || e->cfg.w_bc == 0.f) return GPE_OK;
    int rc = ensure_packed(e);
    if (rc) return rc;
    hipStream_t s0 = e->stream;
    float *g0 = e->grad, *sl0 = e->gslab;
    if (e->side) {
        HIPCHK(e, hipEventRecord(e->ev_fork, s0));
        HIPCHK(e, hipStreamWaitEvent(e->side, e->ev_fork, 0));
        e->stream = e->side;
    }
    e->grad = e->grad_bc;
    if (e->gslab_bc) e->gslab = e->gslab_bc;
    auto body = [&]() -> int {
        int r;
        if ((r = mlp_forward(e, e->bc, true))) return r;
        hipLaunchKernelGGL(k_head_seed_bc, dim3(cdiv(e->bc.n, 256)), dim3(256), 0, e->stream, e->ph, e->base_norm, e->bc.pts.a,
                           e->bc_target, e->bc.O, e->bc.Ob, e->lsums(), e->bc.n, e->bc.ld);
        HIPCHK(e, hipGetLastError());
        if (with_backward && (r = mlp_backward(e, e->bc))) return r;
        return GPE_OK;
    };
    rc = body();
    e->stream = s0; e->grad = g0; e->gslab = sl0;
    if (rc) return rc;
    if (e->side) HIPCHK(e, hipEventRecord(e->ev_join, e->side));
    e->bc_inflight = true;
    return GPE_OK;
}
static int bc_join(gpe_engine* e) {
    if (e->bc_inflight && e->side) HIPCHK(e, hipStreamWaitEvent(e->stream, e->ev_join, 0));
    return GPE_OK;
}
static int launch_tail(gpe_engine* e, bool add_bc) {
    const float* add = (add_bc && e->bc_inflight) ? e->grad_bc : nullptr;
    hipLaunchKernelGGL(k_tail, dim3(add ? cdiv(e->P, 256) : 1), dim3(256), 0, e->stream, e->grad, add, e->P, e->dsc());
    HIPCHK(e, hipGetLastError());
    e->bc_inflight = false;
    return GPE_OK;
}

int gpe_step_backward(gpe_engine* e) {
    if (!e) return GPE_ERR_INVALID;
    if (e->phase != 1) FAIL(e, GPE_ERR_STATE, "step_backward without step_begin");
    int rc;
    const bool with_sym = e->cfg.w_sym != 0.f;
    // small batches of the common problem class (real psi, no orthogonality / Riesz / symmetry terms) on the pipelined reverse
    // kernel: the seeds are formed inside it
    e->seedf_now = seed_in_reverse(e);
    if (!e->seedf_now && (rc = launch_seed_pde(e, nullptr, 1))) return rc;
    rc = mlp_backward(e, e->main, /*close=*/!with_sym);
    e->seedf_now = false;
    e->fh_now = false;
    if (rc) return rc;
    if (with_sym) {
        hipLaunchKernelGGL(k_seed_sym, dim3(cdiv(e->n_pde, 256)), dim3(256), 0, e->stream, e->ph, e->sym.O, e->sym.Ob,
                           e->n_pde, e->sym.ld);
        HIPCHK(e, hipGetLastError());
        if ((rc = mlp_backward(e, e->sym, /*close=*/true, /*assign=*/false))) return rc;
    }
    e->phase = 2;
    return GPE_OK;
}

static void after_update(gpe_engine* e) {          // host-side mirror of what k_update left behind
    e->acc_clean = true;
    e->packed_dirty = e->path == GPE_PATH_FUSED && e->upd_snap != nullptr;      // multi-workgroup update: the next k_begin repacks
}
static void launch_update(gpe_engine* e, const float* grad, const double* sums, const double* lsums, double bc_cnt, int do_update,
                          int mse_mode, double* dbl_keep) {
    const int n_pack = e->path == GPE_PATH_FUSED ? (e->nd.n_lin - 2) * e->H * e->H : 0;
    if (e->upd_snap) {
        hipLaunchKernelGGL(k_update_part, dim3(UPD_G), dim3(1024), 0, e->stream, e->P, grad, sums, lsums, e->od, e->upd_snap);
        hipLaunchKernelGGL(k_update<true>, dim3(UPD_G), dim3(1024), 0, e->stream, e->P, e->theta, e->am, e->av, grad, sums, lsums, e->ph,
                           e->oc, e->od, e->hist, e->cap, e->last, bc_cnt, do_update, mse_mode, e->nd, e->H, e->Wpk, e->WpkT, n_pack, e->dbl,
                           (int)(S_COUNT + LS_COUNT + 4), dbl_keep, (const UpdSnap*)e->upd_snap, 1);
    } else {
        hipLaunchKernelGGL(k_update<false>, dim3(1), dim3(1024), 0, e->stream, e->P, e->theta, e->am, e->av, grad, sums, lsums, e->ph,
                           e->oc, e->od, e->hist, e->cap, e->last, bc_cnt, do_update, mse_mode, e->nd, e->H, e->Wpk, e->WpkT, n_pack, e->dbl,
                           (int)(S_COUNT + LS_COUNT + 4), dbl_keep, (const UpdSnap*)nullptr, ((e->fwd_b6 || e->bwd_b6 || e->H > 64) ? 1 : 2) | (e->update_cache ? 0 : 5));
    }
}
static double bc_count(gpe_engine* e) {
    const int64_t nb = e->nb_merged > 0 ? e->nb_merged : e->bc.n;
    return (nb > 0 && e->cfg.w_bc != 0.f) ? (double)nb * e->nd.n_out : 0.0;
}

int gpe_step_update(gpe_engine* e) {
    if (!e) return GPE_ERR_INVALID;
    if (e->phase != 2) FAIL(e, GPE_ERR_STATE, "step_update without step_backward");
    if (e->fu_done) { e->fu_done = false; after_update(e); e->phase = 0; return GPE_OK; }      // the slab reduction's last workgroup ran it
    if (e->fu_parts) {          // the slab reduction left partial norms + snapshot: the update on UPD_G workgroups, new weights scattered into the packed copies
        e->fu_parts = false;
        const int n_pack = (e->nd.n_lin - 2) * e->H * e->H;
        hipLaunchKernelGGL(k_update<true>, dim3(UPD_G), dim3(1024), 0, e->stream, e->P, e->theta, e->am, e->av, (const float*)e->grad,
                           (const double*)e->sums(), (const double*)e->lsums(), e->ph, e->oc, e->od, e->hist, e->cap, e->last, bc_count(e), 1, 0,
                           e->nd, e->H, e->Wpk, e->WpkT, n_pack, e->dbl, (int)(S_COUNT + LS_COUNT + 4), (double*)nullptr,
                           (const UpdSnap*)e->upd_snap_small, 2);
        HIPCHK(e, hipGetLastError());
        after_update(e);
        e->phase = 0;
        return GPE_OK;
    }
    launch_update(e, e->grad, e->sums(), e->lsums(), bc_count(e), 1, 0, nullptr);
    HIPCHK(e, hipGetLastError());
    after_update(e);
    e->phase = 0;
    return GPE_OK;
}

// ---- pre-training on an analytic target (refine/harmonic_pinn_simulation.py:650-701) ---------------------------------
int gpe_bind_target(gpe_engine* e, const float* d_target) {
    if (!e) return GPE_ERR_INVALID;
    if (e->main.n <= 0) FAIL(e, GPE_ERR_STATE, "bind_target before bind_points");
    e->mse_target = d_target;
    if (!d_target) { free_batch(e->mse); return GPE_OK; }
    return setup_batch(e, e->mse, e->ux, e->n_pde, 1, 0, false, nullptr);
}

int gpe_mse_begin(gpe_engine* e) {
    if (!e) return GPE_ERR_INVALID;
    if (!e->mse_target || e->mse.n <= 0) FAIL(e, GPE_ERR_STATE, "mse step before bind_target");
    int rc;
    e->fh_now = false; e->seedf_now = false;
    e->mse.pts = Pts{e->ux, nullptr, e->n_pde};
    if ((rc = launch_begin(e))) return rc;
    if ((rc = mlp_forward(e, e->mse, true))) return rc;
    hipLaunchKernelGGL(k_seed_mse, dim3(head_grid(e, e->mse.n, 256)), dim3(256), 0, e->stream, e->ph, e->mse.pts.a, e->mse_target,
                       e->mse.O, e->mse.Ob, e->dsc() + 2, e->mse.n, e->mse.ld);
    HIPCHK(e, hipGetLastError());
    if ((rc = mlp_backward(e, e->mse, /*close=*/true))) return rc;
    e->phase = 3;
    return GPE_OK;
}

static int mse_finish(gpe_engine* e, int do_update) {
    if (e->phase != 3) FAIL(e, GPE_ERR_STATE, "mse update without mse begin");
    // sums[S_DEN] must be non-zero for the (unused) Rayleigh quotient of the shared update kernel
    launch_update(e, e->grad, e->sums(), e->lsums(), 0.0, do_update, 1, nullptr);
    HIPCHK(e, hipGetLastError());
    after_update(e);
    e->phase = 0;
    return GPE_OK;
}

int gpe_mse_update(gpe_engine* e) { return e ? mse_finish(e, 1) : GPE_ERR_INVALID; }

int gpe_mse_step(gpe_engine* e, gpe_scalars* out) {
    int rc;
    if ((rc = gpe_mse_begin(e))) return rc;
    if ((rc = mse_finish(e, 1))) return rc;
    if (out) return gpe_read_scalars(e, out);
    return GPE_OK;
}

int gpe_mse_loss_grad(gpe_engine* e, double* loss) {
    int rc;
    if ((rc = gpe_mse_begin(e))) return rc;
    if ((rc = mse_finish(e, 0))) return rc;
    gpe_scalars sc;
    HIPCHK(e, hipMemcpyAsync(&sc, e->last, sizeof sc, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (loss) *loss = sc.loss;
    return GPE_OK;
}

// ---- data-parallel exchange inside the engine -------------------------------------------------------------------
// One process per GPU.  librccl is dlopen'ed here (the copy already mapped into the process -- PyTorch-ROCm's -- if there is one,
// else ROCm's), so libgpe_hip.so has no link-time dependency on it and single-GPU users never load it.
static int rccl_load(gpe_engine* e) {
    if (e->rccl.dl) return GPE_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;      // reuse a mapped copy
    if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) FAIL(e, GPE_ERR_STATE, "librccl.so not found: %s", dlerror());
    e->rccl.dl = h;
    *(void**)&e->rccl.GetUniqueId = dlsym(h, "ncclGetUniqueId");
    *(void**)&e->rccl.CommInitRank = dlsym(h, "ncclCommInitRank");
    *(void**)&e->rccl.CommDestroy = dlsym(h, "ncclCommDestroy");
    *(void**)&e->rccl.AllReduce = dlsym(h, "ncclAllReduce");
    *(void**)&e->rccl.GetErrorString = dlsym(h, "ncclGetErrorString");
    if (!e->rccl.GetUniqueId || !e->rccl.CommInitRank || !e->rccl.CommDestroy || !e->rccl.AllReduce || !e->rccl.GetErrorString) {
        e->rccl.dl = nullptr;
        FAIL(e, GPE_ERR_STATE, "librccl.so lacks a required symbol");
    }
    return GPE_OK;
}
#define RCCLCHK(e, call)                                                                                             \
    do {                                                                                                             \
        ncclResult_t _r = (call);                                                                                    \
        if (_r != ncclSuccess) FAIL(e, GPE_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, (e)->rccl.GetErrorString(_r)); \
    } while (0)

int gpe_comm_unique_id(gpe_engine* e, void* out128) {
    if (!e || !out128) return GPE_ERR_INVALID;
    int rc = rccl_load(e);
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == GPE_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    RCCLCHK(e, e->rccl.GetUniqueId(&id));
    memcpy(out128, &id, sizeof id);
    return GPE_OK;
}

int gpe_comm_init(gpe_engine* e, const void* id128, int rank, int world) {
    if (!e || !id128 || world < 1 || rank < 0 || rank >= world) return GPE_ERR_INVALID;
    if (e->comm) FAIL(e, GPE_ERR_STATE, "communicator already initialised");
    if (world != (e->cfg.world_size > 0 ? e->cfg.world_size : 1))
        FAIL(e, GPE_ERR_INVALID, "comm world %d != gpe_config.world_size %d", world, e->cfg.world_size);
    int rc = rccl_load(e);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    RCCLCHK(e, e->rccl.CommInitRank(&e->comm, world, id, rank));
    e->comm_rank = rank; e->comm_world = world;
    { const char* envi = getenv("GPE_DP_INLINE"); e->dp_inline = !(envi && atoi(envi) == 0); }
    HIPCHK(e, hipStreamCreateWithFlags(&e->comm_stream, hipStreamNonBlocking));
    HIPCHK(e, hipEventCreateWithFlags(&e->ev_x0, hipEventDisableTiming));
    HIPCHK(e, hipEventCreateWithFlags(&e->ev_x1, hipEventDisableTiming));
    e->ev_bucket.resize(e->nd.n_lin);
    for (auto& ev : e->ev_bucket) HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    e->dp_collectives = 0;
    return GPE_OK;
}

int gpe_comm_destroy(gpe_engine* e) {
    if (!e) return GPE_ERR_INVALID;
    if (!e->comm) return GPE_OK;
    (void)hipStreamSynchronize(e->stream);
    (void)hipStreamSynchronize(e->comm_stream);
    (void)e->rccl.CommDestroy(e->comm);
    e->comm = nullptr;
    (void)hipStreamDestroy(e->comm_stream); e->comm_stream = nullptr;
    (void)hipEventDestroy(e->ev_x0); (void)hipEventDestroy(e->ev_x1); e->ev_x0 = e->ev_x1 = nullptr;
    for (hipEvent_t ev : e->ev_bucket) (void)hipEventDestroy(ev);
    e->ev_bucket.clear();
    for (auto& ev : e->ev_g) { if (ev) (void)hipEventDestroy(ev); ev = nullptr; }
    if (e->grad_alt) { (void)hipFree(e->grad_alt); e->grad_alt = nullptr; }
    if (e->dbl_prev) { (void)hipFree(e->dbl_prev); e->dbl_prev = nullptr; }
    e->async_grad = false;
    return GPE_OK;
}

int gpe_comm_info(const gpe_engine* e, int* rank, int* world, int64_t* collectives) {
    if (!e) return GPE_ERR_INVALID;
    if (rank) *rank = e->comm ? e->comm_rank : -1;
    if (world) *world = e->comm ? e->comm_world : 0;
    if (collectives) *collectives = e->dp_collectives;
    return GPE_OK;
}

// all-reduce(sum) buf in place on the exchange stream once everything enqueued on the compute stream so far has finished
static int dp_allreduce_after(gpe_engine* e, void* buf, size_t count, ncclDataType_t dt, hipEvent_t ev) {
    HIPCHK(e, hipEventRecord(ev, e->stream));
    HIPCHK(e, hipStreamWaitEvent(e->comm_stream, ev, 0));
    RCCLCHK(e, e->rccl.AllReduce(buf, buf, count, dt, ncclSum, e->comm, e->comm_stream));
    e->dp_collectives++;
    return GPE_OK;
}
static int dp_allreduce_inline(gpe_engine* e, void* buf, size_t count, ncclDataType_t dt) {     // ... on the compute stream itself
    RCCLCHK(e, e->rccl.AllReduce(buf, buf, count, dt, ncclSum, e->comm, e->stream));
    e->dp_collectives++;
    return GPE_OK;
}
static int dp_join(gpe_engine* e) {       // the compute stream continues after everything on the exchange stream
    HIPCHK(e, hipEventRecord(e->ev_x1, e->comm_stream));
    HIPCHK(e, hipStreamWaitEvent(e->stream, e->ev_x1, 0));
    return GPE_OK;
}

// First phase of an engine-driven data-parallel step: as in gpe_step nobody reads the step sums between the passes, so the forward
// kernel may run the head (the N = 1 kernels: no k_head_pde); its per-workgroup triples are added by one small launch into the sums
// the first all-reduce exchanges, and the reverse phase then reads the GLOBAL sums like after k_head_pde.
static int dp_begin(gpe_engine* e) {
    e->fh_want = true;
    int rc = gpe_step_begin(e);
    e->fh_want = false;
    if (rc) return rc;
    if (e->fh_now) {
        hipLaunchKernelGGL(k_slots_to_sums, dim3(1), dim3(64), 0, e->stream, (const double*)e->head_slots, e->fh_nslots, e->sums(), e->lsums());
        HIPCHK(e, hipGetLastError());
        e->fh_now = false;                          // consumers take the (all-reduced) sums, not the local slots
    }
    return GPE_OK;
}

// One synchronous data-parallel step, no host synchronisation: begin -> all-reduce of the 12 double sums -> backward with the
// gradient all-reduced on the exchange stream (generic set: one bucket per linear map, output map first, behind the remaining
// reverse pass; fused set: the reverse pass is a single kernel, so one P+4 message) -> clip/Adam (replicated, bit-identical).
// One-step-stale variant (gpe_comm_set_async).  Step t: forward / sums exchange / reverse with theta_t -> g_t; its all-reduce is
// issued on the exchange stream and NOT waited for; the update of step t applies g_{t-1} (all-reduced while step t computed) with
// the scalars of step t-1.  Step 0 applies nothing.  Records (gpe_read_scalars, history) therefore lag one step.
static int step_dp_async(gpe_engine* e) {
    int rc;
    const int cur = (int)(e->async_t & 1), prv = cur ^ 1;
    float* gcur = (e->async_t & 1) ? e->grad_alt : e->grad;       // the two gradient buffers alternate
    float* const g_home = e->grad;
    e->grad = gcur;                                   // this step's buffer: zeroed by k_begin (generic set), written by the reverse pass;
                                                      // the other one holds g_{t-1} until this step's update has applied it
    e->acc_clean = e->acc_clean && e->path == GPE_PATH_FUSED;
    rc = dp_begin(e);
    if (!rc) rc = dp_allreduce_after(e, e->sums(), S_COUNT, ncclDouble, e->ev_x0);
    if (!rc) rc = dp_join(e);
    if (!rc) rc = gpe_step_backward(e);
    if (!rc) {
        hipError_t st;
        if ((st = hipEventRecord(e->ev_x0, e->stream)) != hipSuccess || (st = hipStreamWaitEvent(e->comm_stream, e->ev_x0, 0)) != hipSuccess) {
            e->grad = g_home;
            FAIL(e, GPE_ERR_HIP, "stale-gradient exchange: %s", hipGetErrorString(st));
        }
        ncclResult_t nr = e->rccl.AllReduce(gcur, gcur, (size_t)e->P + GT_COUNT, ncclFloat, ncclSum, e->comm, e->comm_stream);
        if (nr != ncclSuccess) { e->grad = g_home; FAIL(e, GPE_ERR_HIP, "ncclAllReduce: %s", e->rccl.GetErrorString(nr)); }
        e->dp_collectives++;
        (void)hipEventRecord(e->ev_g[cur], e->comm_stream);
    }
    e->grad = g_home;
    if (rc) return rc;
    float* gprev = (prv == 1) ? e->grad_alt : e->grad;
    const int apply = e->async_t > 0;
    HIPCHK(e, hipStreamWaitEvent(e->stream, e->ev_g[apply ? prv : cur], 0));     // step 0 only records: it reads its own (reduced) tail
    launch_update(e, apply ? gprev : gcur, apply ? (const double*)e->dbl_prev : (const double*)e->sums(),
                  apply ? (const double*)(e->dbl_prev + S_COUNT) : (const double*)e->lsums(), bc_count(e), apply, 0, e->dbl_prev);
    HIPCHK(e, hipGetLastError());
    after_update(e);
    e->phase = 0;
    e->async_t++;
    return GPE_OK;
}

int gpe_comm_set_async(gpe_engine* e, int on) {
    if (!e) return GPE_ERR_INVALID;
    if (!e->comm) FAIL(e, GPE_ERR_STATE, "gpe_comm_set_async before gpe_comm_init");
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipStreamSynchronize(e->comm_stream));
    if (on && !e->grad_alt) {
        HIPCHK(e, hipMalloc((void**)&e->grad_alt, ((size_t)e->P + GT_COUNT) * sizeof(float) + 256));
        HIPCHK(e, hipMalloc((void**)&e->dbl_prev, (S_COUNT + LS_COUNT + 4) * sizeof(double) + 256));
        HIPCHK(e, hipMemset(e->grad_alt, 0, ((size_t)e->P + GT_COUNT) * sizeof(float)));
        HIPCHK(e, hipMemset(e->dbl_prev, 0, (S_COUNT + LS_COUNT + 4) * sizeof(double)));
        for (auto& ev : e->ev_g) HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    e->async_grad = on != 0;
    e->async_t = 0;
    return GPE_OK;
}

int gpe_step_dp(gpe_engine* e) {
    if (!e) return GPE_ERR_INVALID;
    if (!e->comm) FAIL(e, GPE_ERR_STATE, "gpe_step_dp before gpe_comm_init");
    if (e->async_grad) return step_dp_async(e);
    int rc;
    if ((rc = dp_begin(e))) return rc;
    // layer buckets need every map's gradient to be final when its weight kernel ends: no second batch adding to it later
    const bool bucketed = e->path == GPE_PATH_GENERIC && e->cfg.w_sym == 0.f && !e->bc_inflight;
    // Fused / wide kernel sets: the next kernel needs each all-reduce's result at once and nothing else is enqueued meanwhile, so the
    // collective goes onto the COMPUTE stream itself -- no event hop to the exchange stream and back (each cross-stream dependency
    // costs several microseconds on this runtime: 4 hops were most of the 35-40 us a world-1 step lost against the plain step,
    // profiles/r04/dp_world1_*.json).  The exchange stream stays for what does overlap: the generic set's layer buckets and the
    // opt-in stale-gradient mode.  GPE_DP_INLINE=0 restores the two-stream form.
    const bool inl = !bucketed && e->dp_inline;
    if (inl) rc = dp_allreduce_inline(e, e->sums(), S_COUNT, ncclDouble);
    else if (!(rc = dp_allreduce_after(e, e->sums(), S_COUNT, ncclDouble, e->ev_x0))) rc = dp_join(e);
    if (rc) return rc;
    e->dp_bucket = bucketed;
    rc = gpe_step_backward(e);
    e->dp_bucket = false;
    if (rc) return rc;
    if (bucketed) rc = dp_allreduce_after(e, e->grad + e->P, GT_COUNT, ncclFloat, e->ev_x0);      // exchange tail (sum r^2)
    else if (inl) rc = dp_allreduce_inline(e, e->grad, (size_t)e->P + GT_COUNT, ncclFloat);
    else rc = dp_allreduce_after(e, e->grad, (size_t)e->P + GT_COUNT, ncclFloat, e->ev_x0);
    if (rc) return rc;
    if (!inl && (rc = dp_join(e))) return rc;
    return gpe_step_update(e);
}

int gpe_run_dp(gpe_engine* e, int64_t n_steps) {
    for (int64_t i = 0; i < n_steps; ++i) {
        int rc = gpe_step_dp(e);
        if (rc) return rc;
    }
    return GPE_OK;
}

int gpe_exchange_sums(gpe_engine* e, void** p, int64_t* count) {
    if (!e || !p || !count) return GPE_ERR_INVALID;
    *p = e->sums(); *count = S_COUNT;
    return GPE_OK;
}
int gpe_exchange_grad(gpe_engine* e, void** p, int64_t* count) {
    if (!e || !p || !count) return GPE_ERR_INVALID;
    *p = e->grad; *count = e->P + GT_COUNT;
    return GPE_OK;
}

int gpe_read_scalars(gpe_engine* e, gpe_scalars* out) {
    if (!e || !out) return GPE_ERR_INVALID;
    HIPCHK(e, hipMemcpyAsync(out, e->last, sizeof *out, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (out->nonfinite != 0.0) FAIL(e, GPE_ERR_NONFINITE, "non-finite loss or gradient at step %lld; parameters not updated", (long long)out->step + 1);
    return GPE_OK;
}

int gpe_read_history(gpe_engine* e, int64_t first, int64_t count, gpe_scalars* out) {
    if (!e || !out || first < 1 || count < 0) return GPE_ERR_INVALID;
    if (count > e->cap) FAIL(e, GPE_ERR_INVALID, "history window larger than capacity %d", e->cap);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (int64_t i = 0; i < count;) {
        int64_t slot = (first - 1 + i) % e->cap;
        int64_t run = std::min<int64_t>(count - i, e->cap - slot);
        HIPCHK(e, hipMemcpy(out + i, e->hist + slot, run * sizeof(gpe_scalars), hipMemcpyDeviceToHost));
        i += run;
    }
    return GPE_OK;
}

int gpe_step(gpe_engine* e, gpe_scalars* out) {
    int rc;
    if (!e) return GPE_ERR_INVALID;
    e->fh_want = true;                            // begin and backward are enqueued back to back: the head may ride in the forward kernel
    rc = gpe_step_begin(e);
    e->fh_want = false;
    if (rc) return rc;
    e->fu_want = true;                            // ... and backward and update: the update may ride in the slab reduction
    rc = gpe_step_backward(e);
    e->fu_want = false;
    if (rc) { e->fu_done = false; e->fu_parts = false; return rc; }
    if ((rc = gpe_step_update(e))) return rc;
    if (out) return gpe_read_scalars(e, out);
    return GPE_OK;
}

// Everything a captured step bakes into its kernel nodes: physics and optimiser constants, batch geometry and pointers.
static std::vector<char> graph_key_of(gpe_engine* e) {
    std::vector<char> k;
    auto put = [&](const void* p, size_t n) { const char* c = (const char*)p; k.insert(k.end(), c, c + n); };
    put(&e->ph, sizeof e->ph); put(&e->oc, sizeof e->oc); put(&e->base_norm, sizeof e->base_norm);
    for (Batch* b : {&e->main, &e->bc, &e->sym}) {
        const void* ptrs[] = {b->pts.a, b->pts.b, b->V, b->O, b->Ob, b->u, b->Hu, b->ux, b->stored, b->A0, b->A1};
        put(&b->pts.na, sizeof b->pts.na);
        put(ptrs, sizeof ptrs); put(&b->n, sizeof b->n); put(&b->C, sizeof b->C);
    }
    const void* more[] = {e->bc_target, e->grad, e->dbl, e->stream};
    put(more, sizeof more); put(e->orth_host, sizeof e->orth_host);
    put(&e->cfg.w_bc, sizeof e->cfg.w_bc); put(&e->cfg.w_sym, sizeof e->cfg.w_sym);
    return k;
}

static void graph_drop(gpe_engine* e) {
    if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
    e->graph_key.clear();
}

// capture begin + backward + update on the private capture stream (the caller's stream may be the legacy null stream, which
// cannot be captured); the instantiated graph is launched on the caller's stream.
static int graph_build(gpe_engine* e) {
    graph_drop(e);
    if (!e->cap_stream && hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking) != hipSuccess) return GPE_ERR_HIP;
    std::vector<char> key = graph_key_of(e);
    hipStream_t s0 = e->stream;
    e->stream = e->cap_stream;
    e->packed_dirty = true;                       // the captured step always starts with k_begin
    e->acc_clean = false;
    hipGraph_t g = nullptr;
    int rc = GPE_OK;
    if (hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { e->stream = s0; return GPE_ERR_HIP; }
    // graph_steps consecutive steps per graph: ONE hipGraphLaunch then enqueues all their kernels -- at the reference's batch sizes the
    // host's ~7 us per launch is as long as the kernels themselves (a one-step graph costs more per replay than it saves)
    for (int sidx = 0; sidx < e->graph_steps && !rc; ++sidx) {
        e->fh_want = true;
        rc = gpe_step_begin(e);
        e->fh_want = false;
        if (!rc) { e->fu_want = true; rc = gpe_step_backward(e); e->fu_want = false; }
        if (!rc) rc = gpe_step_update(e);
        e->fu_done = false; e->fu_parts = false;
    }
    hipError_t st = hipStreamEndCapture(e->cap_stream, &g);
    e->stream = s0;
    e->phase = 0;
    e->packed_dirty = true;                       // nothing ran during capture
    e->acc_clean = false;
    if (rc || st != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); return rc ? rc : GPE_ERR_HIP; }
    st = hipGraphInstantiate(&e->graph_exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (st != hipSuccess) { e->graph_exec = nullptr; (void)hipGetLastError(); return GPE_ERR_HIP; }
    e->graph_key.swap(key);
    return GPE_OK;
}

int gpe_run(gpe_engine* e, int64_t n_steps) {
    if (!e) return GPE_ERR_INVALID;
    if (e->use_graph && !e->prof && n_steps >= e->graph_steps && e->main.n > 0 && (e->graph_forced || e->main.n <= e->graph_max_points)) {
        if (!e->graph_exec || e->graph_key != graph_key_of(e)) {
            if (graph_build(e) != GPE_OK) { graph_drop(e); e->use_graph = false; }     // fall back to plain launches for good
        }
        if (e->graph_exec) {
            int64_t done = 0;
            for (; done + e->graph_steps <= n_steps; done += e->graph_steps) HIPCHK(e, hipGraphLaunch(e->graph_exec, e->stream));
            after_update(e);
            e->phase = 0;
            for (; done < n_steps; ++done) {          // remainder: plain launches
                int rc = gpe_step(e, nullptr);
                if (rc) return rc;
            }
            return GPE_OK;
        }
    }
    for (int64_t i = 0; i < n_steps; ++i) {
        int rc = gpe_step(e, nullptr);
        if (rc) return rc;
    }
    return GPE_OK;
}

int gpe_residual(gpe_engine* e, gpe_scalars* out, float* d_psi, float* d_resid) {
    if (!e || !out) return GPE_ERR_INVALID;
    if (e->main.n <= 0) FAIL(e, GPE_ERR_STATE, "residual before bind_points");
    int rc;
    e->fh_now = false; e->seedf_now = false;      // forward-only evaluation: k_head_pde forms the sums, never the slots of an earlier step
    if ((rc = launch_begin(e, /*force=*/true))) return rc;      // no reverse pass here: the gradient buffer must read zero
    if ((rc = mlp_forward(e, e->main, false))) return rc;
    if ((rc = launch_head_pde(e))) return rc;
    if (e->cfg.w_sym != 0.f) {
        if ((rc = mlp_forward(e, e->sym, false))) return rc;
        hipLaunchKernelGGL(k_head_sym, dim3(cdiv(e->n_pde, 256)), dim3(256), 0, e->stream, e->ph, e->sym.O, e->sums(),
                           e->n_pde, e->sym.ld);
    }
    if ((rc = launch_seed_pde(e, d_resid, 0))) return rc;
    if ((rc = bc_fork(e, false))) return rc;
    if ((rc = bc_join(e))) return rc;
    if (d_psi) hipLaunchKernelGGL(k_copy_psi, dim3(cdiv(e->n_pde, 256)), dim3(256), 0, e->stream, e->main.u, d_psi, e->n_pde,
                                  e->main.ld, e->nd.n_out);
    if ((rc = launch_tail(e, false))) return rc;
    launch_update(e, e->grad, e->sums(), e->lsums(), bc_count(e), 0, 0, nullptr);
    HIPCHK(e, hipGetLastError());
    after_update(e);
    e->phase = 0;
    HIPCHK(e, hipMemcpyAsync(out, e->last, sizeof *out, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}

int gpe_set_gamma(gpe_engine* e, float g) { if (!e) return GPE_ERR_INVALID; e->cfg.gamma = g; fill_phys(e); return GPE_OK; }
int gpe_set_power(gpe_engine* e, int p) {
    if (!e) return GPE_ERR_INVALID;
    if (p < 1 || p > 32) FAIL(e, GPE_ERR_INVALID, "power p must be in [1,32]");
    e->cfg.p = p; fill_phys(e); return GPE_OK;
}
int gpe_set_perturb_scale(gpe_engine* e, float s) { if (!e) return GPE_ERR_INVALID; e->cfg.perturb_scale = s; fill_phys(e); return GPE_OK; }
int gpe_set_loss_weights(gpe_engine* e, const float w[6]) {
    if (!e || !w) return GPE_ERR_INVALID;
    if (w[5] != 0.f && e->nd.n_out != 1 && !(e->cfg.complex_psi && e->nd.n_out == 2 && e->cfg.p == 3))
        FAIL(e, GPE_ERR_INVALID, "the Riesz energy term needs real psi (out=1) or complex psi (out=2) with p = 3");
    if ((w[3] != 0.f) != (e->cfg.w_sym != 0.f)) FAIL(e, GPE_ERR_INVALID, "the symmetry term cannot be switched on/off after bind_points");
    e->cfg.w_pde = w[0]; e->cfg.w_bc = w[1]; e->cfg.w_norm = w[2]; e->cfg.w_sym = w[3]; e->cfg.w_orth = w[4]; e->cfg.w_riesz = w[5];
    fill_phys(e);
    return GPE_OK;
}
int gpe_set_n_global(gpe_engine* e, int64_t n) { if (!e) return GPE_ERR_INVALID; e->cfg.n_global = n; fill_phys(e); return GPE_OK; }
int gpe_set_lr(gpe_engine* e, float lr) {
    if (!e) return GPE_ERR_INVALID;
    OptDev h;
    HIPCHK(e, hipMemcpyAsync(&h, e->od, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    h.lr = lr; h.lr0 = lr;
    HIPCHK(e, hipMemcpyAsync(e->od, &h, sizeof h, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return GPE_OK;
}

int gpe_profile_enable(gpe_engine* e, int on) {
    if (!e) return GPE_ERR_INVALID;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->prof = on != 0;
    e->ev_used = 0;
    return GPE_OK;
}
int gpe_profile_read(gpe_engine* e, double out[4]) {
    if (!e || !out) return GPE_ERR_INVALID;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    out[0] = out[1] = out[2] = out[3] = 0.0;
    for (size_t i = 0; i < e->ev_used; ++i) {
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->ev_pool[2 * i], e->ev_pool[2 * i + 1]));
        out[e->ev_kind[i] ? 2 : 0] += ms;
        out[e->ev_kind[i] ? 3 : 1] += 1.0;
    }
    e->ev_used = 0;
    return GPE_OK;
}

int gpe_stop_state(gpe_engine* e, int* stopped, int64_t* stop_step) {
    if (!e) return GPE_ERR_INVALID;
    OptDev h;
    HIPCHK(e, hipMemcpyAsync(&h, e->od, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (stopped) *stopped = h.stopped;
    if (stop_step) *stop_step = h.stop_step;
    return GPE_OK;
}

// diagnostic builds (-DGPE_STAMP) only: read and clear the per-phase cycle counters of the reverse kernel
int gpe_debug_read_stamps(gpe_engine* e, unsigned long long out[16]) {
    if (!e || !out) return GPE_ERR_INVALID;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 16 * sizeof(unsigned long long)));
    unsigned long long z[16] = {0};
    HIPCHK(e, hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z));
    return GPE_OK;
}

#ifdef GPE_STAMP
// tuning builds only (not part of include/gpe_hip.h): the per-workgroup phase trace of f_backward_pipe
extern "C" int gpe_debug_read_trace(gpe_engine* e, unsigned long long* out, int n) {
    if (!e || !out || n > 512 * 4 * 32) return GPE_ERR_INVALID;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), (size_t)n * sizeof(unsigned long long)));
    return GPE_OK;
}
#endif

int gpe_step_cost(const gpe_engine* e, double* flops_per_point, double* hbm_bytes_per_point) {
    if (!e) return GPE_ERR_INVALID;
    // SURVEY 8(d) with the channel count of the training batches, C = d + 2 (value, d first derivatives, Laplacian) instead of
    // 1 + 2d:  F_fwd = 2 d H + C [2 H^2 (L-1) + 2 H out] (+ activations), F_step = 3 F_fwd ; B_mat = 2 C H L 4
    const int d = e->nd.dim, L = e->nd.n_lin - 1, H = e->nd.width[1], no = e->nd.n_out;
    const double C = d + 2;
    double gemm = 2.0 * d * H + C * (2.0 * H * H * (L - 1) + 2.0 * H * no);
    double act = (double)H * L * (6 + 4 * d);
    if (flops_per_point) *flops_per_point = 3.0 * (gemm + act);
    if (hbm_bytes_per_point) *hbm_bytes_per_point = 2.0 * C * H * L * 4.0;
    return GPE_OK;
}

}  // extern "C"
