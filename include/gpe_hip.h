/*
 * gpe_hip.h -- C ABI of libgpe_hip.so: the MI355X (gfx950) engine for the Gross-Pitaevskii
 * eigenvalue-residual PINN training step.
 *
 * The reference (LevBahn/Gross-Pitaevskii-Eigenvalue-problem) has no FFI for this path: it is a set of
 * Python methods on torch tensors.  Each entry point below names the reference code it replaces
 * (paths relative to /root/reference; "refine/" = Gross-Pitaevskii/src/final/refine/,
 * "nb cN:Lm" = Gross_Pitaevskii_1D_power_Test.ipynb code cell N, line m).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  d_* pointers are DEVICE pointers
 *     (e.g. tensor.data_ptr() of a PyTorch-ROCm tensor used as storage); h_* are host pointers.
 *   - Every function returns GPE_OK (0) or a negative gpe_status; gpe_last_error() gives the text.
 *     No exception crosses the ABI.
 *   - A handle is single-device and not thread-safe.  One process per GPU.  Data-parallel exchange: either
 *     by the engine itself -- gpe_comm_unique_id / gpe_comm_init create an RCCL communicator and gpe_step_dp /
 *     gpe_run_dp issue both all-reduces of a step on a dedicated stream -- or by the caller, on the two device
 *     buffers exposed by gpe_exchange_sums() / gpe_exchange_grad() between the three phases of a step.
 *   - All kernels run on the stream given at gpe_create() (NULL = the legacy default stream).
 *   - Parameters are one flat fp32 vector in torch state_dict order:
 *       network.0.weight [out,in] row-major, network.0.bias, network.2.weight, ...
 *     (refine/harmonic_pinn_simulation.py:84-93).
 *   - Derivative "jets": channel 0 = value, 1..d = d/dx_k, d+1..2d = d2/dx_k^2 (C = 1+2d channels).
 */
#ifndef GPE_HIP_H
#define GPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPE_ABI_VERSION 2
#define GPE_MAX_LAYERS 12 /* entries of layers[]: [d, H_1, ..., H_L, out] */
#define GPE_MAX_ORTH 4
#define GPE_MAX_DIM 3

typedef enum gpe_status {
    GPE_OK = 0,
    GPE_ERR_INVALID = -1,     /* bad argument / unsupported configuration (ValueError in the reference: refine/harmonic_pinn_simulation.py:143) */
    GPE_ERR_HIP = -2,         /* HIP runtime error */
    GPE_ERR_NONFINITE = -3,   /* loss or gradient not finite; parameters were NOT updated */
    GPE_ERR_STATE = -4,       /* call sequence error (e.g. step before bind_points) */
    GPE_ERR_NOMEM = -5
} gpe_status;

enum { GPE_ACT_TANH = 0, GPE_ACT_TANH_PLUS1 = 1 };                         /* nb c6:L38 ; refine/harmonic_pinn_simulation.py:41-49 */
enum { GPE_POT_HARMONIC = 0, GPE_POT_GAUSSIAN = 1, GPE_POT_PERIODIC = 2,   /* nb c6:L63-79 ; refine/...:136-144 */
       GPE_POT_PRECOMPUTED = 3, GPE_POT_NONE = 4 };
enum { GPE_SCHED_CONST = 0, GPE_SCHED_COSINE_LOSS = 1, GPE_SCHED_PLATEAU = 2 }; /* refine/...:312-314,361 (quirk Q4) ; nb c10:L76-78,L103 */
enum { GPE_PATH_AUTO = 0, GPE_PATH_GENERIC = 1, GPE_PATH_FUSED = 2 };
/* analytic base phi_n of the perturbation ansatz u = phi_n + s*NN: Hermite (refine/harmonic_pinn_simulation.py:95-119),
 * box sine sqrt(2/L) sin((n+1) pi x / L) (refine/box_pinn_simulation.py:99-117), or three caller-supplied arrays
 * (phi, phi', phi'' on the bound points: e.g. the Airy base of refine/gravity_well_pinn_simulation.py:97-173) */
enum { GPE_BASE_HERMITE = 0, GPE_BASE_BOX = 1, GPE_BASE_PRECOMPUTED = 2 };
/* hard boundary factor multiplying the network output in model.forward: none, or sin(pi x / env_L)
 * (refine/box_pinn_simulation.py:119-130) */
enum { GPE_ENV_NONE = 0, GPE_ENV_SIN = 1 };
enum { GPE_RIESZ_PAPER = 0, GPE_RIESZ_SUM = 1, GPE_RIESZ_VARIATIONAL = 2 };
enum { GPE_NET_MLP = 0, GPE_NET_RESIDUAL = 1 };
/* eigenvalue estimate inside the residual:  RAYLEIGH  lambda = sum u (H u) / sum u^2  (refine/harmonic_pinn_simulation.py:186-188; its
 * branch of the gradient vanishes identically);  ENERGY  lambda = [c sum |grad u|^2 + sum V u^2 + gamma sum |u|^(p+1)] / sum u^2,
 * the energy-functional form of the 2D classes (src/gross_pitaevskii_2D.py:192, src/gross_pitaevskii_2D_minimal.py:179; per-point
 * reading of quirk Q1), whose branch of the gradient does NOT vanish and is carried through the reverse pass */
enum { GPE_LAMBDA_RAYLEIGH = 0, GPE_LAMBDA_ENERGY = 1 };

typedef struct gpe_engine gpe_engine; /* opaque */

/* Problem + optimiser description.  Mirrors the literals of refine/harmonic_pinn_simulation.py:963-1009 and
 * nb c20/c22, and the constructor arguments of GrossPitaevskiiPINN (refine/...:57 ; nb c6:L6). */
typedef struct gpe_config {
    int32_t abi_version;          /* = GPE_ABI_VERSION */
    int32_t n_layers;             /* entries used in layers[] */
    int32_t layers[GPE_MAX_LAYERS];
    int32_t activation;           /* GPE_ACT_* */
    int32_t complex_psi;          /* 1: layers[last]==2, psi = out0 + i*out1 */
    float kinetic_coeff;          /* c in -c*laplacian: 1 (refine/...:181) or 0.5 (nb c6:L113) */
    int32_t potential;            /* GPE_POT_* */
    float pot_scale;              /* harmonic: V = pot_scale * sum_k (omega[k]*(x_k - c_k))^2, c = (pot_a, 0, 0): the beta-scaled shifted trap
                                   * V = beta/2 omega^2 (x - center)^2 of refine/vary_potential_parameter_harmonic.py:231-240 is
                                   * pot_scale = beta/2, omega[0] = omega, pot_a = center */
    float omega[GPE_MAX_DIM];
    float pot_a, pot_v0, pot_k;   /* gaussian centre (and harmonic trap centre along x); periodic depth, wave number */
    float omega_rot;              /* rotation frequency Omega: -Omega*L_z psi (complex psi, dim>=2) */
    float gamma;                  /* interaction strength (refine/...:184) */
    int32_t p;                    /* nonlinearity power: gamma*u^p */
    int32_t abs_power;            /* 1: gamma*|u|^(p-1)*u */
    int32_t base_mode;            /* -1 none; n>=0: u = phi_n(x) + perturbation (refine/...:127-134) */
    int32_t base_deriv;           /* 0 exact base derivatives; 1 notebook quirk Q8 (H_n constant, nb c6:L45-47) */
    float perturb_scale;          /* q/normal_const (refine/...:333-340); 1 for the notebook surface */
    float bc_nn_scale;            /* NN scale inside boundary_loss (quirk Q7: 1.0, refine/...:202-206) */
    float w_pde, w_bc, w_norm, w_sym, w_orth;   /* refine/...:347,355 ; nb c10:L97 */
    float sym_sign;               /* +1 even mode, -1 odd mode (nb c6:L150-153) */
    float dx;                     /* quadrature weight of normalization_loss (refine/...:212-217) */
    int64_t n_global;             /* N of the means, over ALL ranks (0: use the bound local count) */
    /* optimiser: torch.optim.Adam defaults + clip_grad_norm_ (refine/...:309,359-360) */
    float lr, beta1, beta2, eps, clip_norm;   /* clip_norm <= 0: no clipping */
    int32_t sched;                /* GPE_SCHED_* */
    float T_0, T_mult, eta_min;   /* cosine warm restarts (refine/...:312-314) */
    float factor; int32_t patience; float min_lr, threshold;   /* ReduceLROnPlateau (nb c10:L76-78) */
    /* execution */
    int32_t path;                 /* GPE_PATH_* */
    int32_t world_size;           /* data-parallel ranks sharing the replicated boundary batch (>=1) */
    int32_t history_capacity;     /* steps of scalar history kept on device (0: default 65536) */
    /* early stopping, evaluated on the device after every update (refine/...:363-400): stop when loss <= stop_tol or
     * after stop_patience steps without a new best loss; once stopped, further steps leave the parameters untouched.
     * stop_tol <= 0 and stop_patience <= 0 disable the respective test. */
    float stop_tol;
    int32_t stop_patience;
    /* row f3 of SURVEY 8: other bases / hard boundary factor (1D) */
    int32_t base_kind;            /* GPE_BASE_* (used when base_mode >= 0) */
    int32_t envelope;             /* GPE_ENV_* */
    float box_L;                  /* L of the box base */
    float env_L;                  /* L of the sin(pi x / L) factor */
    /* row f4: Riesz energy term  w_riesz * E  (real psi, any dimension), riesz_kind =
     *   GPE_RIESZ_PAPER       E = [1/2 sum |grad u|^2 + sum V u^2 + gamma/(p+1) sum |u|^(p+1)] / sum u^2
     *                         (Notebooks/Paper/Gross_Pitaevskii_1D_Harmonic.ipynb c6:L133-183)
     *   GPE_RIESZ_SUM         E = 1/2 sum |grad u|^2 + 1/2 sum V u^2 + gamma/(p+1) sum |u|^(p+1)   (unnormalised point sums:
     *                         src/gross_pitaevskii_2D.py:112-151, p = 3: 1/2 (K + P + eta/2 sum u^4))
     *   GPE_RIESZ_VARIATIONAL E = energy of the NORMALISED state u/sqrt(I), I = dx sum u^2:
     *                         [c sum |grad u|^2 + sum V u^2] / sum u^2 + 2 gamma/(p+1) sum |u|^(p+1) / (sum u^2 * I^((p-1)/2)).
     *                         Scale-invariant; its stationary points are exactly the solutions of
     *                         -c lap v + V v + gamma |v|^(p-1) v = mu v, int v^2 = 1, and its minimiser is the ground state -- the
     *                         term that keeps training off the excited states without biasing the other loss terms */
    float w_riesz;
    int32_t riesz_kind;
    /* row f3: network flavour.  GPE_NET_RESIDUAL: layers = [d, H, ..., H, out] describes Linear(d,H) + activation, len(layers)-3
     * residual blocks  tanh(lin2(tanh(lin1 x)) + x)  and Linear(H,out) (refine/box_to_gaussian_pinn_simulation.py:52-63,100-130);
     * parameters in state_dict order network.0, network.2.lin1, network.2.lin2, network.3.lin1, ... (generic kernel set) */
    int32_t net_kind;
    int32_t lambda_kind;          /* GPE_LAMBDA_* (ENERGY: real psi, odd p) */
    /* regularisers of the 2D classes' pde_loss (src/gross_pitaevskii_2D.py:197-211 ; arXiv 2010.05075), added to the total loss:
     *   w_reg_f   / (mean(u^2) + reg_f_eps)      "L_f": keeps the network off the trivial eigenfunction (reference: w = 1, eps = 1e-2)
     *   w_reg_lam / (lambda^2 + reg_lam_eps)     "L_lambda": keeps lambda off zero (reference: w = 1, eps = 1e-6; needs GPE_LAMBDA_ENERGY)
     * mean over n_global; both zero: the terms are not formed */
    float w_reg_f, reg_f_eps, w_reg_lam, reg_lam_eps;
} gpe_config;

/* Per-step scalars (refine/...:364-381 keeps loss every 10 and lambda every 100 epochs). */
typedef struct gpe_scalars {
    double loss, pde, bc, norm, sym, orth;
    double mu;                    /* Rayleigh quotient lambda_pde (refine/...:186-188) */
    double num, den, sum_r2, integral;
    double grad_norm, lr;
    double step;                  /* 1-based optimiser step that produced this record */
    double nonfinite;             /* 1 when the loss or gradient was not finite (no update) */
    double riesz;                 /* Riesz energy E (0 when w_riesz == 0) */
    double reg;                   /* w_reg_f / (mean u^2 + eps) + w_reg_lam / (lambda^2 + eps)  (0 when both weights are 0) */
} gpe_scalars;

/* ---- lifetime ------------------------------------------------------------------------------ */
int gpe_abi_version(void);
size_t gpe_sizeof_config(void);   /* sizeof(gpe_config)  -- lets a foreign-language binding verify its struct layout */
size_t gpe_sizeof_scalars(void);  /* sizeof(gpe_scalars) */
/* replaces: GrossPitaevskiiPINN(...).to(device) + torch.optim.Adam(...) + scheduler construction
 * (refine/harmonic_pinn_simulation.py:295,309-314 ; nb c10:L63,L73-78) */
int gpe_create(const gpe_config* cfg, int device, void* hip_stream, gpe_engine** out);
void gpe_destroy(gpe_engine* e);
const char* gpe_last_error(const gpe_engine* e); /* e may be NULL: last create() error */
/* 1 = generic kernels, 2 = fused MFMA kernels (what gpe_create selected) */
int gpe_active_path(const gpe_engine* e);
/* "fwd=<kernel>;bwd=<kernel>": the jet forward / reverse kernels the bound collocation batch is dispatched to (the engine
 * picks among several by shape, batch size and the GPE_* tuning switches).  Needs gpe_bind_points first. */
int gpe_active_kernels(gpe_engine* e, char* buf, size_t n);

/* ---- parameters: model.state_dict() / load_state_dict() (refine/...:299,909,953) --------------- */
/* Hidden widths: the MFMA kernel sets exist for one width of 32, 64, 128 or 256 (>= 2 hidden layers).  An MLP with other hidden widths <= 256 is
 * run padded to the next of those with units that output exactly 0 (zero weights; bias 0 for tanh, -40 for tanh + 1): the same function and
 * gradients, and the padding does not move under Adam.  Every call below takes and returns the network AS GIVEN in gpe_config.layers.
 * Residual blocks, widths above 256 and single-hidden-layer networks run on the generic layer-by-layer kernel set (widths above 256 padded to a
 * multiple of 256 for its MFMA kernels). */
int64_t gpe_param_count(const gpe_engine* e);
int gpe_set_params(gpe_engine* e, const float* h_flat, size_t n);
int gpe_get_params(gpe_engine* e, float* h_flat, size_t n);
int gpe_get_grad(gpe_engine* e, float* h_flat, size_t n);            /* gradient of the last step (after exchange, before clipping) */
int gpe_get_adam_state(gpe_engine* e, float* h_m, float* h_v, size_t n, int64_t* step);
int gpe_set_adam_state(gpe_engine* e, const float* h_m, const float* h_v, size_t n, int64_t step);
int gpe_reset_optimizer(gpe_engine* e, float lr);                    /* new Adam + scheduler state (refine/...:309-314 per gamma) */

/* ---- data binding: X_tensor, boundary_points (refine/...:260-264) ----------------------------------- */
/* d_x [n_local, dim] row-major fp32; d_V [n_local] or NULL (precomputed_potential, refine/...:146,175-178).
 * Buffers stay owned by the caller and must outlive their use. */
int gpe_bind_points(gpe_engine* e, const float* d_x, int64_t n_local, const float* d_V);
int gpe_bind_boundary(gpe_engine* e, const float* d_xb, int64_t n_b, const float* d_target /* [n_b,out] or NULL = 0 */);
int gpe_bind_orth(gpe_engine* e, int k, const float* d_psi_k /* [n_local] or NULL to clear */);
/* GPE_BASE_PRECOMPUTED: phi, phi', phi'' of the base on the bound points, each [n_local].  The boundary term then uses no
 * base: fold phi(x_b) into the boundary target. */
int gpe_bind_base(gpe_engine* e, const float* d_phi, const float* d_phi1, const float* d_phi2);

/* ---- forward-only entry points ---------------------------------------------------------------- */
/* model.forward(x) (refine/...:121-125): d_out [n, out] row-major */
int gpe_forward(gpe_engine* e, const float* d_x, int64_t n, float* d_out);
/* NN output jets [C][n][out] (replaces the two torch.autograd.grad calls of refine/...:158-172 at the NN output) */
int gpe_forward_jets(gpe_engine* e, const float* d_x, int64_t n, float* d_jets);
/* pde_loss (refine/...:146-196 ; nb c6:L81-127) + boundary/normalisation/symmetry terms on the bound data,
 * no parameter update.  d_psi [n_local,out], d_residual [n_local,out] may be NULL.  Single-rank semantics. */
int gpe_residual(gpe_engine* e, gpe_scalars* out, float* d_psi, float* d_residual);
/* plot_wavefunction normalisation (refine/...:463-474 ; nb c12:L30-42): u = base + scale*NN on the grid d_x,
 * u /= sqrt(sum u^2 * dx), |u| if abs_flag; d_u [n,out], d_density [n] (either may be NULL) */
int gpe_eval_density(gpe_engine* e, const float* d_x, int64_t n, float dx, int abs_flag, float* d_u, float* d_density);

/* ---- training step: the epoch body refine/...:328-361 ; nb c10:L84-103 ------------------------------ */
/* phase 1: forward jets + local sums  -> exchange buffer "sums" */
int gpe_step_begin(gpe_engine* e);
/* phase 2: residual, seeds, reverse pass -> exchange buffer "grad" (flat gradient + tail scalars) */
int gpe_step_backward(gpe_engine* e);
/* phase 3: clip_grad_norm_, Adam, scheduler.step(loss), history record */
int gpe_step_update(gpe_engine* e);
/* device buffers to all-reduce(sum) between the phases when world_size > 1 */
int gpe_exchange_sums(gpe_engine* e, void** d_ptr /* double* */, int64_t* count);
int gpe_exchange_grad(gpe_engine* e, void** d_ptr /* float*  */, int64_t* count);
/* Optional: make the engine use CALLER-owned device memory for the two exchange buffers, so that they can be
 * handed to a collective library as that library's own tensor type.  d_dbl: >= gpe_exchange_dbl_count() doubles
 * (the first `count` of gpe_exchange_sums() are the all-reduced part), d_grad: >= P + 4 floats. */
int64_t gpe_exchange_dbl_count(void);
int gpe_use_external_exchange(gpe_engine* e, void* d_dbl, int64_t n_dbl, void* d_grad, int64_t n_grad);
/* ---- data-parallel exchange INSIDE the engine: RCCL over xGMI on a dedicated HIP stream ------------------------------------
 * The reference is single-device (refine/...:12); this is the build's row (e) of SURVEY 8.  One process per GPU:
 *   rank 0: gpe_comm_unique_id() -> 128 bytes, handed to the other ranks by any out-of-band channel (a torch.distributed
 *   store, MPI, a file); every rank: gpe_comm_init(id, rank, world) with gpe_config.world_size == world.
 * gpe_step_dp = gpe_step_begin, ncclAllReduce of the 12 double sums, gpe_step_backward with the gradient all-reduced on the
 * exchange stream (generic kernel set: one bucket per linear map, output map first, overlapped with the rest of the reverse
 * pass; fused set: one P+4 message), gpe_step_update.  Nothing synchronises with the host. */
#define GPE_COMM_ID_BYTES 128
int gpe_comm_unique_id(gpe_engine* e, void* out128);
int gpe_comm_init(gpe_engine* e, const void* id128, int rank, int world);
int gpe_comm_destroy(gpe_engine* e);
int gpe_comm_info(const gpe_engine* e, int* rank, int* world, int64_t* collectives);   /* rank -1 / world 0: no communicator */
int gpe_step_dp(gpe_engine* e);
int gpe_run_dp(gpe_engine* e, int64_t n_steps);
/* OPT-IN one-step-stale gradient: the all-reduce of step t's gradient stays in flight on the exchange stream behind the forward
 * of step t+1; the update of step t applies the (all-reduced) gradient and scalars of step t-1, step 0 applies nothing, records lag
 * one step.  It changes the optimisation trajectory -- not comparable bit for bit with the reference -- and is never enabled
 * implicitly; a run that uses it must say so. */
int gpe_comm_set_async(gpe_engine* e, int on);
/* all three phases + synchronise + scalars (single rank) */
int gpe_step(gpe_engine* e, gpe_scalars* out);
/* n steps enqueued back to back, no host synchronisation (single rank) */
int gpe_run(gpe_engine* e, int64_t n_steps);
/* synchronise and read the scalars of the most recent step / of steps [first, first+count) (1-based) */
int gpe_read_scalars(gpe_engine* e, gpe_scalars* out);
int gpe_read_history(gpe_engine* e, int64_t first_step, int64_t count, gpe_scalars* out);
int gpe_synchronize(gpe_engine* e);
/* early-stop state: *stopped = 1 once a stop condition fired; *stop_step = the optimiser step that fired it (1-based) */
int gpe_stop_state(gpe_engine* e, int* stopped, int64_t* stop_step);

/* ---- pre-training on an analytic target: pretrain_on_analytical_solution (refine/...:650-701) ------------------------
 * loss = mean((NN(x) - target)^2) on the bound points; d_target [n_local,out].  gpe_mse_step: one plain Adam step (no
 * clipping, no scheduler, refine/...:663-670); gpe_mse_loss_grad: loss + gradient without update (the reference's L-BFGS
 * tail, refine/...:672-687, runs host-side on these; read the gradient with gpe_get_grad).  Single-rank entry points; the
 * phases gpe_mse_begin / gpe_mse_update bracket an all-reduce of the gradient exchange buffer when world_size > 1. */
int gpe_bind_target(gpe_engine* e, const float* d_target);
int gpe_mse_begin(gpe_engine* e);                    /* forward, seeds, reverse -> gradient exchange buffer (tail: sum of squares) */
int gpe_mse_update(gpe_engine* e);                   /* Adam on the exchanged gradient */
int gpe_mse_step(gpe_engine* e, gpe_scalars* out);   /* begin + update + synchronise; out->loss = the MSE */
int gpe_mse_loss_grad(gpe_engine* e, double* loss);

/* ---- continuation knobs: the gamma / perturbation loop of refine/...:289-340 -------------------------- */
int gpe_set_gamma(gpe_engine* e, float gamma);
int gpe_set_power(gpe_engine* e, int p);
int gpe_set_lr(gpe_engine* e, float lr);
int gpe_set_perturb_scale(gpe_engine* e, float s);
int gpe_set_n_global(gpe_engine* e, int64_t n_global);
/* loss weights between steps: the host-side ReLoBRaLo balancing of src/gross_pitaevskii_2D_ReLoBRaLo.py:259-342 only needs the
 * per-term scalars gpe_residual / gpe_step return and this setter.  w[6] = {pde, bc, norm, sym, orth, riesz}. */
int gpe_set_loss_weights(gpe_engine* e, const float w[6]);

/* Per-kernel timing with HIP events on the engine's stream (for bench.py's roofline object).  While enabled, every
 * launch of the two dominant kernels on the collocation batch (jet forward, jet reverse) is bracketed by events.
 * gpe_profile_read synchronises and returns out[0]=forward ms total, out[1]=forward launches, out[2]=reverse ms total,
 * out[3]=reverse launches (since the last enable), then clears the counters. */
int gpe_profile_enable(gpe_engine* e, int on);
int gpe_profile_read(gpe_engine* e, double out[4]);

/* bytes of HBM traffic one step is designed to move (B_mat-style accounting, for bench.py) and FLOPs */
int gpe_step_cost(const gpe_engine* e, double* flops_per_point, double* hbm_bytes_per_point);

#ifdef __cplusplus
}
#endif
#endif /* GPE_HIP_H */
