"""Shared test helpers: golden-fixture loading and fixture -> oracle Problem mapping."""
import glob
import os

import numpy as np

from oracle import gpe_oracle as go

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fx(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def refine_names():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "fx_refine_*.npz")))


def nb_names():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "fx_nb_*.npz")))


def problem_from_refine(fx) -> go.Problem:
    """refine/harmonic_pinn_simulation.py flavour: ShiftedTanh, -u'' + x^2 u + gamma u^p, 10*bc + 20*norm."""
    return go.Problem(layers=[int(v) for v in fx["layers"]], activation=1, kinetic_coeff=1.0,
                      potential=go.POT_HARMONIC, pot_scale=1.0, gamma=float(fx["gamma"]), p=int(fx["p"]),
                      base_mode=int(fx["mode"]), base_deriv=0,
                      perturb_scale=float(fx["perturb_const"]) / float(fx["normal_const"]), bc_nn_scale=1.0,
                      w_bc=10.0, w_norm=20.0, w_sym=0.0, dx=float(fx["dx"]))


def problem_from_nb(fx) -> go.Problem:
    """root-notebook flavour: tanh, -1/2 u'' + 1/2 x^2 u + gamma u^p, 10*bc + 20*norm + 5*sym."""
    mode = int(fx["mode"])
    return go.Problem(layers=[int(v) for v in fx["layers"]], activation=0, kinetic_coeff=0.5,
                      potential=go.POT_HARMONIC, pot_scale=0.5, gamma=float(fx["gamma"]), p=int(fx["p"]),
                      base_mode=mode, base_deriv=1, perturb_scale=1.0, bc_nn_scale=1.0,
                      w_bc=10.0, w_norm=20.0, w_sym=5.0, sym_sign=(-1.0 if mode % 2 == 1 else 1.0),
                      dx=float(fx["dx"]))


def problem_from_vanilla(fx) -> go.Problem:
    """use_perturbation=False of refine/harmonic_pinn_simulation.py (:152-155, :205-208): the scaled network output IS u (no base) in the
    residual and the norm; the boundary term takes the raw network output."""
    return go.Problem(layers=[int(v) for v in fx["layers"]], activation=1, kinetic_coeff=1.0, potential=go.POT_HARMONIC, pot_scale=1.0,
                      gamma=float(fx["gamma"]), p=int(fx["p"]), base_mode=-1,
                      perturb_scale=float(fx["perturb_const"]) / float(fx["normal_const"]), bc_nn_scale=1.0,
                      w_bc=10.0, w_norm=20.0, w_sym=0.0, dx=float(fx["dx"]))


def bc_points(fx):
    return np.array([[float(fx["lb"])], [float(fx["ub"])]], dtype=np.float64)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def problem_from_box(fx, mode=None, const=None) -> go.Problem:
    """refine/box_pinn_simulation.py flavour: ShiftedTanh, -u'' + gamma u^p on [0,1], forward = NN*sin(pi x), sine base."""
    mode = int(fx["mode"]) if mode is None else mode
    const = float(fx["normal_const"]) if const is None else const
    return go.Problem(layers=[int(v) for v in fx["layers"]], activation=1, kinetic_coeff=1.0, potential=go.POT_NONE,
                      gamma=float(fx["gamma"]) if "gamma" in fx else 0.0, p=int(fx["p"]) if "p" in fx else 3,
                      base_mode=mode, base_kind=go.BASE_BOX, box_L=1.0, envelope=go.ENV_SIN, env_L=1.0,
                      perturb_scale=float(fx["perturb_const"]) / const, bc_nn_scale=1.0, w_bc=10.0, w_norm=20.0,
                      dx=float(fx["dx"]) if "dx" in fx else 1.0 / (int(fx["N"]) - 1))


def problem_from_gravity(fx) -> go.Problem:
    """refine/gravity_well_pinn_simulation.py flavour: V = x (fed as a precomputed potential), Airy base fed as arrays."""
    return go.Problem(layers=[int(v) for v in fx["layers"]], activation=1, kinetic_coeff=1.0, potential=go.POT_PRECOMPUTED,
                      gamma=float(fx["gamma"]), p=int(fx["p"]), base_mode=int(fx["mode"]), base_kind=go.BASE_PRECOMPUTED,
                      perturb_scale=float(fx["perturb_const"]) / float(fx["normal_const"]), bc_nn_scale=1.0,
                      w_bc=10.0, w_norm=20.0, dx=float(fx["dx"]))


def problem_from_paper(fx) -> go.Problem:
    """Notebooks/Paper/Gross_Pitaevskii_1D_Harmonic.ipynb flavour, mode 0: Riesz + PDE + 10 bc + 20 norm + 5 sym."""
    return go.Problem(layers=[int(v) for v in fx["layers"]], activation=0, kinetic_coeff=0.5, potential=go.POT_HARMONIC,
                      pot_scale=0.5, gamma=float(fx["gamma"]), p=int(fx["p"]), abs_power=True, base_mode=0, base_deriv=1,
                      perturb_scale=1.0, w_bc=10.0, w_norm=20.0, w_sym=5.0, w_riesz=1.0, dx=float(fx["dx"]))


def problem_from_box2gauss(fx) -> go.Problem:
    """refine/box_to_gaussian_pinn_simulation.py flavour: residual-block network, sine base on [0, ub], V = exp(-(x - 0.5)^2),
    -u'' + V u + gamma u^p, forward = network(x) (no boundary factor), 10*bc + 20*norm."""
    return go.Problem(layers=[int(v) for v in fx["layers"]], net_kind=go.NET_RESIDUAL, activation=1, kinetic_coeff=1.0,
                      potential=go.POT_GAUSSIAN, pot_a=0.5, gamma=float(fx["gamma"]), p=int(fx["p"]), base_mode=int(fx["mode"]),
                      base_kind=go.BASE_BOX, box_L=float(fx["ub"]), perturb_scale=float(fx["perturb_const"]) / float(fx["normal_const"]),
                      bc_nn_scale=1.0, w_bc=10.0, w_norm=20.0, dx=float(fx["dx"]))


VBETA_SEEDS = {"fx_vbeta_harmonic_m0_b0.4_g0.npz": 0, "fx_vbeta_harmonic_m3_b1_g2_p4.npz": 2, "fx_vbeta_gravity_m0_b0.5_g0.npz": 0,
               "fx_vbeta_gravity_m1_b2_g5.npz": 1, "fx_vbeta_boxgauss_m0_b10_g0.npz": 0, "fx_vbeta_boxgauss_m1_b3_g5_p2.npz": 1}


def vbeta_names():
    return sorted(VBETA_SEEDS)


def problem_from_vbeta(fx):
    """The beta-sweep flavours (refine/vary_potential_parameter_{harmonic,gravity_well,box_and_gaussian}.py; tests/golden/
    make_golden_vary_beta.py) -> (Problem, dict of the arrays the oracle / engine are handed: V_pre, base_pre, bc_target).
    harmonic : box [0, L], sine base, V = beta/2 omega^2 (x - 2.5)^2 with omega = 10, formed analytically (pot_scale = beta/2, pot_a)
    gravity  : Airy base as arrays, potential term beta * x * u -> precomputed potential beta x
    boxgauss : forward = NN sin(pi x), sine base, potential term beta exp(-x^2/2) u -> precomputed potential"""
    flavour = str(fx["flavour"])
    beta = float(fx["beta"])
    x = fx["x"].astype(np.float64)
    common = dict(layers=[int(v) for v in fx["layers"]], activation=1, kinetic_coeff=1.0, gamma=float(fx["gamma"]), p=int(fx["p"]),
                  base_mode=int(fx["mode"]), perturb_scale=float(fx["perturb_const"]) / float(fx["normal_const"]), bc_nn_scale=1.0,
                  w_bc=10.0, w_norm=20.0, dx=float(fx["dx"]))
    if flavour == "harmonic":
        return go.Problem(potential=go.POT_HARMONIC, pot_scale=0.5 * beta, omega=(10.0, 1.0, 1.0), pot_a=2.5, base_kind=go.BASE_BOX,
                          box_L=float(fx["ub"]), **common), {}
    if flavour == "gravity":
        return (go.Problem(potential=go.POT_PRECOMPUTED, base_kind=go.BASE_PRECOMPUTED, **common),
                dict(V_pre=beta * x[:, 0], base_pre=(fx["base"][:, 0], fx["base_x"][:, 0], fx["base_xx"][:, 0]),
                     bc_target=-fx["base_boundary"]))
    return (go.Problem(potential=go.POT_PRECOMPUTED, base_kind=go.BASE_BOX, box_L=float(fx["ub"]), envelope=go.ENV_SIN, env_L=1.0,
                       **common), dict(V_pre=beta * np.exp(-x[:, 0] ** 2 / 2)))


def stage_divergence(pb, start_flat, X, xb, hist, K, sched, lr, **opt):
    """Engine history `hist` (list of per-epoch records of one driver stage) against the fp64 oracle stepped K epochs from the SAME
    start weights with the same optimiser settings: per-epoch relative loss error and absolute mu error (arrays of length K).  The
    oracle trajectory is the yardstick; the divergence grows with k (Adam + clipping amplify rounding), which the callers bound."""
    st = go.OptState(lr0=float(lr), sched=sched, **opt)
    _, tr = go.train_steps(pb, st, np.asarray(start_flat, np.float64), np.asarray(X, np.float64), K, np.asarray(xb, np.float64), dtype=np.float64)
    K = min(K, len(hist))
    dl = np.array([abs(hist[k]["loss"] - tr[k]["loss"]) / max(abs(tr[k]["loss"]), 1e-30) for k in range(K)])
    dm = np.array([abs(hist[k]["mu"] - tr[k]["mu"]) for k in range(K)])
    dlr = np.array([abs(hist[k]["lr"] - tr[k]["lr"]) for k in range(K)])
    return dl, dm, dlr


# ---- the 2D classes' loss (tests/golden/make_golden_2d_class.py) --------------------------------------------------------------------
CLASS2D = ["fx_2d_class_32x2_g100.npz", "fx_2d_class_64x4_g500.npz", "fx_2d_class_100x3_g100.npz"]


def gaussian_2d(x, V0=1.0, x0=np.pi / 2, y0=np.pi / 2, sigma=0.5):
    """compute_potential of the 2D classes (src/gross_pitaevskii_2D.py:244-274)"""
    return V0 * np.exp(-((x[:, 0] - x0) ** 2 + (x[:, 1] - y0) ** 2) / (2 * sigma ** 2))


def problem_from_class2d(fx, n_global=0) -> go.Problem:
    """src/gross_pitaevskii_2D.py:215-242: 10 mean(u_bc^2) + Riesz sum + mean(r^2) + 1/(mean u^2 + 1e-2) + 1/(lambda^2 + 1e-6), energy-functional lambda"""
    return go.Problem(layers=[int(v) for v in fx["layers"]], activation=0, kinetic_coeff=1.0, potential=go.POT_PRECOMPUTED, gamma=float(fx["g"]), p=3,
                      abs_power=True, w_pde=1.0, w_bc=10.0, w_norm=0.0, w_riesz=1.0, riesz_kind=go.RIESZ_SUM, lambda_kind=go.LAMBDA_ENERGY,
                      w_reg_f=1.0, reg_f_eps=1e-2, w_reg_lam=1.0, reg_lam_eps=1e-6, dx=1.0, n_global=n_global)
