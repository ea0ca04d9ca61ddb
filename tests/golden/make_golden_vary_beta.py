#!/usr/bin/env python3
"""
tests/golden/make_golden_vary_beta.py -- golden vectors for the beta-SWEEP flavours of the refine family (VERDICT r03
"Missing 3"), produced by importing the reference's own scripts (read-only):
  /root/reference/Gross-Pitaevskii/src/final/refine/vary_potential_parameter_harmonic.py          (class :52-342, driver :344-556)
  /root/reference/Gross-Pitaevskii/src/final/refine/vary_potential_parameter_gravity_well.py      (class :52-257, driver :259-470)
  /root/reference/Gross-Pitaevskii/src/final/refine/vary_potential_parameter_box_and_gaussian.py  (class :52-225, driver :227-439)
Each: GrossPitaevskiiPINN(layers, ..., beta, [L]), compute_potential, pde_loss(inputs, predictions, gamma, beta, p, ...), the epoch-0
body of train_gpe_model(gamma, beta_values, ...) -> op-level tensors, loss terms and the gradient (fx_vbeta_<flavour>_*.npz); and
one short seeded run of each driver's beta continuation (fx_vbetadriver_<flavour>.npz).  Arrays only; runs only in the build container.

Usage:  MPLBACKEND=Agg python tests/golden/make_golden_vary_beta.py
"""
import contextlib
import importlib.util
import io
import os
import time

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np
import torch

REF = "/root/reference/Gross-Pitaevskii/src/final/refine"
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)

FLAVOURS = {   # name: (script, potential_type, (lb, ub), constructor takes L)
    "harmonic": ("vary_potential_parameter_harmonic.py", "harmonic", (0.0, 5.0), True),
    "gravity": ("vary_potential_parameter_gravity_well.py", "gravity_well", (0.0, 10.0), False),
    "boxgauss": ("vary_potential_parameter_box_and_gaussian.py", "gaussian", (0.0, 1.0), True),
}


def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def flat_params(model):
    return np.concatenate([p.detach().numpy().ravel() for p in model.parameters()]).astype(np.float32)


def make_model(mod, flavour, layers, mode, beta, ub):
    if FLAVOURS[flavour][3]:
        return mod.GrossPitaevskiiPINN(layers, mode=mode, beta=beta, L=ub)
    return mod.GrossPitaevskiiPINN(layers, mode=mode, beta=beta)


def oplevel(mod, flavour, tag, layers, N, seed, mode, beta, gamma, p, perturb_const=0.01):
    """epoch-0 body of train_gpe_model (harmonic :446-474) at a given beta, from advanced_initialization weights"""
    _, pot, (lb, ub), _ = FLAVOURS[flavour]
    torch.manual_seed(seed)
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    dx = X[1, 0] - X[0, 0]
    model = make_model(mod, flavour, layers, mode, beta, ub)
    model.apply(lambda m: mod.advanced_initialization(m, mode))
    flat0 = flat_params(model)
    X_tensor = torch.tensor(X, dtype=torch.float32, requires_grad=True)
    bpts = torch.tensor([[lb], [ub]], dtype=torch.float32)
    bvals = torch.zeros((2, 1), dtype=torch.float32)
    u_nn = model.forward(X_tensor)
    normal_const = torch.max(u_nn).detach().clone()
    u_pred = perturb_const * (u_nn / normal_const)
    u = model.get_complete_solution(X_tensor, u_pred)
    extra = {}
    if flavour == "gravity":      # the Airy base as this class builds it (scipy on the host, phi'' by np.gradient: :135-173) and at the two boundary points
        b0, b1, b2 = model.get_complete_solution_with_derivatives(X_tensor, u_pred * 0.0)
        extra = dict(base=b0.detach().numpy(), base_x=b1.detach().numpy(), base_xx=b2.detach().numpy(),
                     base_boundary=model.airy_solution(bpts, mode).detach().numpy())
    pde_loss, lam = model.pde_loss(X_tensor, u_pred, gamma, beta, p, pot)
    bl = model.boundary_loss(bpts, bvals)
    nl = model.normalization_loss(model.get_complete_solution(X_tensor, u_pred), dx)
    total = pde_loss + 10.0 * bl + 20.0 * nl
    model.zero_grad()
    total.backward()
    grad0 = np.concatenate([q.grad.detach().numpy().ravel() for q in model.parameters()]).astype(np.float32)
    V = (model.compute_potential(X_tensor, beta, pot) if flavour == "harmonic" else model.compute_potential(X_tensor, pot)).detach().numpy()
    fx = dict(flavour=flavour, layers=np.array(layers), N=N, seed=seed, mode=mode, beta=beta, gamma=gamma, p=p,
              perturb_const=perturb_const, normal_const=float(normal_const), dx=dx, lb=lb, ub=ub, flat0=flat0, x=X.astype(np.float32),
              forward_out=u_nn.detach().numpy(), u=u.detach().numpy(), V=V, lam=float(lam), pde_loss=float(pde_loss),
              bc_loss=float(bl), norm_loss=float(nl), total=float(total), grad0=grad0)
    fx.update(extra)
    np.savez_compressed(os.path.join(OUT, f"fx_vbeta_{flavour}_{tag}.npz"), **fx)
    print("wrote vbeta", flavour, tag, "loss0 %.6g lam0 %.6g" % (fx["total"], fx["lam"]), flush=True)


def driver(mod, flavour, layers, N, seed, mode, betas, gamma, p, epochs, tol, lr=1e-3, perturb_const=0.01):
    """the reference's own train_gpe_model(gamma, beta_values, ...): pre-training at beta = 0, warm-started beta continuation"""
    _, pot, (lb, ub), _ = FLAVOURS[flavour]
    torch.manual_seed(seed)
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        models, lam_table, hist, const, ep = mod.train_gpe_model(gamma, betas, [mode], p, X, lb, ub, layers, epochs, tol, perturb_const,
                                                                potential_type=pot, lr=lr, verbose=False)
    wall = time.time() - t0
    fx = dict(flavour=flavour, layers=np.array(layers), N=N, seed=seed, mode=mode, betas=np.array(betas, dtype=np.float64), gamma=gamma,
              p=p, epochs=epochs, tol=tol, lr=lr, perturb_const=perturb_const, lb=lb, ub=ub, wall_seconds=wall,
              lam_table=np.array(lam_table[mode], dtype=np.float64), const=float(const[mode]),
              stop_epochs=np.array([ep[mode][b] for b in betas], dtype=np.int64))
    for b in betas:
        h = hist[mode][b]
        fx[f"loss_b{b}"] = np.array(h["loss"], dtype=np.float64)
        fx[f"lambda_b{b}"] = np.array(h["lambda"], dtype=np.float64)
        fx[f"constraint_b{b}"] = np.array(h["constraint"], dtype=np.float64)
        fx[f"flat_b{b}"] = flat_params(models[mode][b])
    np.savez_compressed(os.path.join(OUT, f"fx_vbetadriver_{flavour}.npz"), **fx)
    print("wrote vbetadriver", flavour, lam_table[mode], ep[mode], f"{wall:.0f} s", flush=True)


if __name__ == "__main__":
    mods = {k: load_module(f"ref_vbeta_{k}", os.path.join(REF, v[0])) for k, v in FLAVOURS.items()}
    oplevel(mods["harmonic"], "harmonic", "m0_b0.4_g0", [1, 64, 64, 64, 1], 400, 0, 0, 0.4, 0.0, 3)
    oplevel(mods["harmonic"], "harmonic", "m3_b1_g2_p4", [1, 32, 32, 32, 32, 1], 300, 2, 3, 1.0, 2.0, 4)      # mode >= 3: the small-gain initialisation
    oplevel(mods["gravity"], "gravity", "m0_b0.5_g0", [1, 64, 64, 64, 1], 400, 0, 0, 0.5, 0.0, 3)
    oplevel(mods["gravity"], "gravity", "m1_b2_g5", [1, 32, 32, 32, 1], 300, 1, 1, 2.0, 5.0, 3)
    oplevel(mods["boxgauss"], "boxgauss", "m0_b10_g0", [1, 64, 64, 64, 1], 400, 0, 0, 10.0, 0.0, 3)
    oplevel(mods["boxgauss"], "boxgauss", "m1_b3_g5_p2", [1, 32, 32, 32, 1], 300, 1, 1, 3.0, 5.0, 2)
    driver(mods["harmonic"], "harmonic", [1, 64, 64, 64, 1], 300, 0, 0, [0.0, 0.05, 0.1], 0.0, 3, 400, 1e-5)
    driver(mods["gravity"], "gravity", [1, 32, 32, 32, 1], 300, 1, 0, [1.0, 2.0, 3.0], 0.0, 3, 400, 1e-5)      # (this script takes normal_const at beta == 1: :374)
    driver(mods["boxgauss"], "boxgauss", [1, 32, 32, 32, 1], 300, 2, 0, [0.0, 5.0, 10.0], 1.0, 3, 400, 1e-5)
