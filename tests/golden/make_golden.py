#!/usr/bin/env python3
"""
tests/golden/make_golden.py -- generate the golden vectors under tests/golden/ by IMPORTING THE
REFERENCE ITSELF (read-only, from /root/reference) and running its own classes on seeded inputs.

Runs only in the build container (the reference does not travel to the GPU box); the .npz files it
writes are committed.  Nothing under tests/ reads /root/reference at test time.

What is imported:
  refine = Gross-Pitaevskii/src/final/refine/harmonic_pinn_simulation.py   (module import; __main__ guarded)
  nb     = Gross_Pitaevskii_1D_power_Test.ipynb code cells c6 (class), c10 (train_gpe_model), c18 (init)
           exec'd into a namespace (cells c3/c4 -- pip/git, optional optimisers -- are NOT run)
  g2d    = Gross-Pitaevskii/src/gross_pitaevskii_2D_minimal.py            (only to document quirk Q1)

Usage:  MPLBACKEND=Agg python tests/golden/make_golden.py
"""
import importlib.util
import io
import json
import math
import os
import sys
import contextlib
import warnings

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_notebook_ns():
    nbj = json.load(open(os.path.join(REF, "Gross_Pitaevskii_1D_power_Test.ipynb")))
    from scipy.special import hermite
    ns = dict(torch=torch, nn=nn, np=np, math=math, hermite=hermite, device=torch.device("cpu"), os=os)
    for ci in (6, 10, 18):
        exec("".join(nbj["cells"][ci]["source"]), ns)
    return ns


def flat_params(model):
    return np.concatenate([p.detach().numpy().ravel() for p in model.parameters()]).astype(np.float32)


def flat_grads(model):
    return np.concatenate([p.grad.detach().numpy().ravel() for p in model.parameters()]).astype(np.float32)


# ------------------------------------------------------------------------------------------------
def refine_fixture(refine, tag, layers, N, seed, mode, gamma, p, perturb_const, n_trace):
    """refine/ PL-PINN flavour: ShiftedTanh, -u'' + x^2 u, cosine scheduler stepped with the loss."""
    torch.manual_seed(seed)
    lb, ub = -10.0, 10.0
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    dx = X[1, 0] - X[0, 0]
    model = refine.GrossPitaevskiiPINN(layers, mode=mode, gamma=gamma)
    model.apply(lambda m: refine.advanced_initialization(m, mode))
    flat0 = flat_params(model)
    X_tensor = torch.tensor(X, dtype=torch.float32, requires_grad=True)
    bpts = torch.tensor([[lb], [ub]], dtype=torch.float32)
    bvals = torch.zeros((2, 1), dtype=torch.float32)

    # ---- op-level tensors at the initial weights (epoch-0 body, harmonic_pinn_simulation.py:332-355)
    u_nn = model.forward(X_tensor)
    normal_const = torch.max(u_nn).detach().clone()
    u_pred = perturb_const * (u_nn / normal_const)
    u = model.get_complete_solution(X_tensor, u_pred)
    u_x = torch.autograd.grad(u, X_tensor, torch.ones_like(u), create_graph=True, retain_graph=True)[0]
    u_xx = torch.autograd.grad(u_x, X_tensor, torch.ones_like(u_x), create_graph=True, retain_graph=True)[0]
    V = model.compute_potential(X_tensor, "harmonic")
    pde_loss, lam = model.pde_loss(X_tensor, u_pred, gamma, p, "harmonic")
    bl = model.boundary_loss(bpts, bvals)
    nl = model.normalization_loss(model.get_complete_solution(X_tensor, u_pred), dx)
    total = pde_loss + 10.0 * bl + 20.0 * nl
    model.zero_grad()
    total.backward()
    grad0 = flat_grads(model)
    with torch.no_grad():
        Hu = -u_xx + V * u + gamma * u ** p
        resid = Hu - lam * u

    fx = dict(layers=np.array(layers), N=N, seed=seed, mode=mode, gamma=gamma, p=p,
              perturb_const=perturb_const, normal_const=float(normal_const), dx=dx, lb=lb, ub=ub,
              flat0=flat0, x=X.astype(np.float32), nn_out=u_nn.detach().numpy(),
              u=u.detach().numpy(), u_x=u_x.detach().numpy(), u_xx=u_xx.detach().numpy(),
              V=V.detach().numpy(), lam=float(lam), residual=resid.numpy(), pde_loss=float(pde_loss),
              bc_loss=float(bl), norm_loss=float(nl), total=float(total), grad0=grad0)

    # ---- n_trace epochs of the real loop body (:328-361) with the reference's optimiser objects
    from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    scheduler = CosineAnnealingWarmRestarts(optimizer, T_0=200, T_mult=2, eta_min=1e-6)
    tr = dict(loss=[], mu=[], lr=[], gnorm=[], pde=[], bc=[], norm=[])
    snaps = {}
    for epoch in range(n_trace):
        optimizer.zero_grad()
        u_pred = model.forward(X_tensor)
        u_pred = perturb_const * u_pred
        u_pred = u_pred / normal_const
        bl = model.boundary_loss(bpts, bvals)
        nl = model.normalization_loss(model.get_complete_solution(X_tensor, u_pred), dx)
        pde_loss, lam = model.pde_loss(X_tensor, u_pred, gamma, p, "harmonic")
        total = pde_loss + (10.0 * bl + 20.0 * nl)
        total.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        tr['lr'].append(optimizer.param_groups[0]['lr'])
        optimizer.step()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scheduler.step(total)
        tr['loss'].append(float(total)); tr['mu'].append(float(lam)); tr['gnorm'].append(float(gn))
        tr['pde'].append(float(pde_loss)); tr['bc'].append(float(bl)); tr['norm'].append(float(nl))
        if epoch in (0, 1, 2):
            snaps[f"flat_after_{epoch + 1}"] = flat_params(model)
    for k, v in tr.items():
        fx["trace_" + k] = np.array(v, dtype=np.float64)
    fx.update(snaps)
    fx["flat_final"] = flat_params(model)
    # eval path (:463-474)
    X_test = np.linspace(lb, ub, 1000).reshape(-1, 1)
    with torch.no_grad():
        Xt = torch.tensor(X_test, dtype=torch.float32)
        up = model.forward(Xt) * (perturb_const / normal_const)
        full_u = model.get_complete_solution(Xt, up)
        u_np = full_u.cpu().numpy().flatten()
        u_np /= np.sqrt(np.sum(u_np ** 2) * (X_test[1, 0] - X_test[0, 0]))
        if mode == 0:
            u_np = np.abs(u_np)
    fx["eval_x"] = X_test.astype(np.float32)
    fx["eval_u"] = u_np
    np.savez_compressed(os.path.join(OUT, f"fx_refine_{tag}.npz"), **fx)
    print("wrote", tag, "loss0", fx['total'], "lam0", fx['lam'], "loss_end", tr['loss'][-1])


# ------------------------------------------------------------------------------------------------
def notebook_fixture(ns, tag, layers, N, seed, mode, gamma, power, n_trace):
    """root-notebook flavour: tanh, -1/2 u'' + 1/2 x^2 u, + 5*sym, ReduceLROnPlateau."""
    torch.manual_seed(seed)
    lb, ub = -10.0, 10.0
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    dx = X[1, 0] - X[0, 0]
    model = ns['GrossPitaevskiiPINN'](layers, mode=mode, power=power)
    model.apply(lambda m: ns['advanced_initialization'](m, mode))
    flat0 = flat_params(model)
    X_tensor = torch.tensor(X, dtype=torch.float32, requires_grad=True)
    bpts = torch.tensor([[lb], [ub]], dtype=torch.float32)
    bvals = torch.zeros((2, 1), dtype=torch.float32)

    u_pred = model.forward(X_tensor)
    pde_loss, resid, lam, full_u = model.pde_loss(power, X_tensor, u_pred, gamma, "harmonic")
    bl = model.boundary_loss(bpts, bvals)
    nl = model.normalization_loss(full_u, dx)
    sl = model.symmetry_loss(X_tensor, lb, ub)
    total = pde_loss + 10.0 * bl + 20.0 * nl + 5.0 * sl
    model.zero_grad()
    total.backward()
    grad0 = flat_grads(model)
    fx = dict(layers=np.array(layers), N=N, seed=seed, mode=mode, gamma=gamma, p=power, dx=dx, lb=lb, ub=ub,
              flat0=flat0, x=X.astype(np.float32), nn_out=u_pred.detach().numpy(), u=full_u.detach().numpy(),
              residual=resid.detach().numpy(), lam=float(lam), pde_loss=float(pde_loss), bc_loss=float(bl),
              norm_loss=float(nl), sym_loss=float(sl), total=float(total), grad0=grad0)

    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=0.5, patience=100,
                                                           min_lr=1e-5)
    tr = dict(loss=[], mu=[], lr=[], gnorm=[])
    for epoch in range(n_trace):
        optimizer.zero_grad()
        u_pred = model.forward(X_tensor)
        pde_loss, _, lam, full_u = model.pde_loss(power, X_tensor, u_pred, gamma, "harmonic")
        bl = model.boundary_loss(bpts, bvals)
        nl = model.normalization_loss(full_u, dx)
        sl = model.symmetry_loss(X_tensor, lb, ub)
        total = pde_loss + 10.0 * bl + 20.0 * nl + 5.0 * sl
        total.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        tr['lr'].append(optimizer.param_groups[0]['lr'])
        optimizer.step()
        scheduler.step(total)
        tr['loss'].append(float(total)); tr['mu'].append(float(lam)); tr['gnorm'].append(float(gn))
        if epoch in (0, 1, 2):
            fx[f"flat_after_{epoch + 1}"] = flat_params(model)
    for k, v in tr.items():
        fx["trace_" + k] = np.array(v, dtype=np.float64)
    fx["flat_final"] = flat_params(model)
    # density on the test grid (notebook c12:L30-42)
    X_test = np.linspace(lb, ub, 1000).reshape(-1, 1)
    with torch.no_grad():
        Xt = torch.tensor(X_test, dtype=torch.float32)
        full = model.get_complete_solution(Xt, model.forward(Xt), mode)
        u_np = full.numpy().flatten()
        u_np /= np.sqrt(np.sum(u_np ** 2) * (X_test[1, 0] - X_test[0, 0]))
    fx["eval_x"] = X_test.astype(np.float32)
    fx["eval_density"] = u_np ** 2
    np.savez_compressed(os.path.join(OUT, f"fx_nb_{tag}.npz"), **fx)
    print("wrote nb", tag, "loss0", fx['total'], "lam0", fx['lam'], "loss_end", tr['loss'][-1])


def notebook_driver_fixture(ns, tag, layers, N, seed, epochs):
    """Run the notebook's OWN train_gpe_model (c10) end to end for a short schedule."""
    torch.manual_seed(seed)
    lb, ub = -10, 10
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    buf = io.StringIO()
    # torch>=2.7 dropped ReduceLROnPlateau(verbose=...), which cell c10:L76-78 still passes (the reference pins
    # torch 2.4.1): accept-and-ignore that one keyword for the duration of the call.
    orig = torch.optim.lr_scheduler.ReduceLROnPlateau

    class _Plateau(orig):
        def __init__(self, *a, verbose=None, **k):
            super().__init__(*a, **k)

    torch.optim.lr_scheduler.ReduceLROnPlateau = _Plateau
    try:
        with contextlib.redirect_stdout(buf), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            models, mu_table = ns['train_gpe_model']([1], [2, 3], [0, 1], X, lb, ub, layers, epochs,
                                                     potential_type='harmonic', lr=1e-3, verbose=False)
    finally:
        torch.optim.lr_scheduler.ReduceLROnPlateau = orig
    fx = dict(layers=np.array(layers), N=N, seed=seed, epochs=epochs, gamma=1.0)
    for mode, logs in mu_table.items():
        fx[f"mu_mode{mode}"] = np.array(logs, dtype=np.float64)       # rows (power, final_mu)
        for power, m in models[mode].items():
            fx[f"flat_mode{mode}_p{power}"] = flat_params(m)
    np.savez_compressed(os.path.join(OUT, f"fx_nbdriver_{tag}.npz"), **fx)
    print("wrote nbdriver", tag, {k: v.tolist() for k, v in fx.items() if k.startswith('mu_')})


def quirk_q1_fixture(g2d):
    """Document quirk Q1: the 2D reference residual is [N,N] (src/gross_pitaevskii_2D_minimal.py:170-182)."""
    torch.manual_seed(0)
    model = g2d.GrossPitaevskiiPINN([2, 8, 8, 1], g=1.0)
    x = torch.rand(7, 2, requires_grad=True)
    out = model.pde_loss(x, model.forward(x))
    resid = out[1]
    np.savez_compressed(os.path.join(OUT, "fx_q1_2d_shape.npz"), residual_shape=np.array(resid.shape))
    print("Q1 residual shape", tuple(resid.shape))


def schedule_fixture():
    """Quirk Q4 closed form: lr produced by CosineAnnealingWarmRestarts.step(loss) for a sweep of loss values."""
    from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts
    losses = np.concatenate([np.logspace(-6, 4, 61), [199.999, 200.0, 200.5, 599.9, 600.0, 601.0, 1400.0, 5797.0]])
    lrs = []
    for L in losses:
        opt = torch.optim.Adam([torch.zeros(1, requires_grad=True)], lr=1e-3)
        s = CosineAnnealingWarmRestarts(opt, T_0=200, T_mult=2, eta_min=1e-6)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            s.step(torch.tensor(float(L)))
        lrs.append(opt.param_groups[0]['lr'])
    np.savez_compressed(os.path.join(OUT, "fx_cosine_loss_lr.npz"), loss=losses, lr=np.array(lrs))
    print("wrote cosine sweep")


if __name__ == "__main__":
    refine = load_module("ref_refine_harmonic",
                         os.path.join(REF, "Gross-Pitaevskii/src/final/refine/harmonic_pinn_simulation.py"))
    ns = load_notebook_ns()
    g2d = load_module("ref_g2d_minimal", os.path.join(REF, "Gross-Pitaevskii/src/gross_pitaevskii_2D_minimal.py"))

    refine_fixture(refine, "m0_g0_64x3", [1, 64, 64, 64, 1], 512, 0, 0, 0.0, 3, 0.01, 60)
    refine_fixture(refine, "m0_g50_64x3", [1, 64, 64, 64, 1], 512, 1, 0, 50.0, 3, 0.01, 60)
    refine_fixture(refine, "m2_g10_32x4", [1, 32, 32, 32, 32, 1], 384, 2, 2, 10.0, 3, 0.01, 40)
    refine_fixture(refine, "m5_g4_p4_64x4", [1, 64, 64, 64, 64, 1], 256, 3, 5, 4.0, 4, 0.01, 20)
    notebook_fixture(ns, "m0_g1_p3_32x4", [1, 32, 32, 32, 32, 1], 512, 0, 0, 1.0, 3, 60)
    notebook_fixture(ns, "m0_g100_p3_64x4", [1, 64, 64, 64, 64, 1], 512, 1, 0, 100.0, 3, 40)
    notebook_fixture(ns, "m0_g1_p2_64x3", [1, 64, 64, 64, 1], 400, 2, 0, 1.0, 2, 40)
    notebook_driver_fixture(ns, "small", [1, 32, 32, 1], 256, 7, 201)
    quirk_q1_fixture(g2d)
    schedule_fixture()


# ------------------------------------------------------------------------------------------------
def box_fixture(box, tag, layers, N, seed, mode, gamma, p, perturb_const):
    """refine/box_pinn_simulation.py flavour (row f3): forward = NN * sin(pi x), base sqrt(2/L) sin((n+1) pi x / L), V = 0."""
    torch.manual_seed(seed)
    lb, ub = 0.0, 1.0
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    dx = X[1, 0] - X[0, 0]
    model = box.GrossPitaevskiiPINN(layers, mode=mode, gamma=gamma, L=1.0)
    model.apply(lambda m: box.advanced_initialization(m, mode))
    flat0 = flat_params(model)
    X_tensor = torch.tensor(X, dtype=torch.float32, requires_grad=True)
    bpts = torch.tensor([[lb], [ub]], dtype=torch.float32)
    bvals = torch.zeros((2, 1), dtype=torch.float32)
    u_nn = model.forward(X_tensor)
    normal_const = torch.max(u_nn).detach().clone()
    u_pred = perturb_const * (u_nn / normal_const)
    u, u_x, u_xx = model.get_complete_solution_with_derivatives(X_tensor, u_pred)
    pde_loss, lam = model.pde_loss(X_tensor, u_pred, gamma, p, "box")
    bl = model.boundary_loss(bpts, bvals)
    nl = model.normalization_loss(model.get_complete_solution(X_tensor, u_pred), dx)
    total = pde_loss + 10.0 * bl + 20.0 * nl
    model.zero_grad()
    total.backward()
    fx = dict(layers=np.array(layers), N=N, seed=seed, mode=mode, gamma=gamma, p=p, perturb_const=perturb_const,
              normal_const=float(normal_const), dx=dx, lb=lb, ub=ub, flat0=flat0, x=X.astype(np.float32),
              forward_out=u_nn.detach().numpy(), u=u.detach().numpy(), u_x=u_x.detach().numpy(), u_xx=u_xx.detach().numpy(),
              lam=float(lam), pde_loss=float(pde_loss), bc_loss=float(bl), norm_loss=float(nl), total=float(total),
              grad0=flat_grads(model))
    np.savez_compressed(os.path.join(OUT, f"fx_box_{tag}.npz"), **fx)
    print("wrote box", tag, "loss0", fx['total'], "lam0", fx['lam'])


if __name__ == "__main__" and os.environ.get("GOLDEN_BOX", "1") == "1":
    box = load_module("ref_refine_box", os.path.join(REF, "Gross-Pitaevskii/src/final/refine/box_pinn_simulation.py"))
    box_fixture(box, "m0_g0", [1, 64, 64, 64, 1], 400, 0, 0, 0.0, 3, 0.01)
    box_fixture(box, "m1_g20", [1, 64, 64, 64, 1], 400, 5, 1, 20.0, 3, 0.01)


# ------------------------------------------------------------------------------------------------
def gravity_fixture(gw, tag, layers, N, seed, mode, gamma, p, perturb_const):
    """refine/gravity_well_pinn_simulation.py flavour (row f3): V = x on [0,35], Airy base (scipy, host), phi'' by np.gradient."""
    torch.manual_seed(seed)
    lb, ub = 0.0, 35.0
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    dx = X[1, 0] - X[0, 0]
    model = gw.GrossPitaevskiiPINN(layers, mode=mode, gamma=gamma)
    model.apply(lambda m: gw.advanced_initialization(m, mode))
    flat0 = flat_params(model)
    X_tensor = torch.tensor(X, dtype=torch.float32, requires_grad=True)
    bpts = torch.tensor([[lb], [ub]], dtype=torch.float32)
    bvals = torch.zeros((2, 1), dtype=torch.float32)
    u_nn = model.forward(X_tensor)
    normal_const = torch.max(u_nn).detach().clone()
    u_pred = perturb_const * (u_nn / normal_const)
    u, u_x, u_xx = model.get_complete_solution_with_derivatives(X_tensor, u_pred)
    with torch.no_grad():
        zero = torch.zeros_like(u_pred)
    b0, b1, b2 = model.get_complete_solution_with_derivatives(X_tensor, u_pred * 0.0)
    pde_loss, lam = model.pde_loss(X_tensor, u_pred, gamma, p, "gravity_well")
    bl = model.boundary_loss(bpts, bvals)
    nl = model.normalization_loss(model.get_complete_solution(X_tensor, u_pred), dx)
    total = pde_loss + 10.0 * bl + 20.0 * nl
    model.zero_grad()
    total.backward()
    base_b = model.airy_solution(bpts, mode).detach().numpy()
    fx = dict(layers=np.array(layers), N=N, seed=seed, mode=mode, gamma=gamma, p=p, perturb_const=perturb_const,
              normal_const=float(normal_const), dx=dx, lb=lb, ub=ub, flat0=flat0, x=X.astype(np.float32),
              base=b0.detach().numpy(), base_x=b1.detach().numpy(), base_xx=b2.detach().numpy(), base_boundary=base_b,
              u=u.detach().numpy(), u_x=u_x.detach().numpy(), u_xx=u_xx.detach().numpy(),
              lam=float(lam), pde_loss=float(pde_loss), bc_loss=float(bl), norm_loss=float(nl), total=float(total),
              grad0=flat_grads(model))
    np.savez_compressed(os.path.join(OUT, f"fx_gravity_{tag}.npz"), **fx)
    print("wrote gravity", tag, "loss0", fx['total'], "lam0", fx['lam'], "normal_const", fx['normal_const'])


if __name__ == "__main__" and os.environ.get("GOLDEN_GRAVITY", "1") == "1":
    gw = load_module("ref_refine_gravity", os.path.join(REF, "Gross-Pitaevskii/src/final/refine/gravity_well_pinn_simulation.py"))
    gravity_fixture(gw, "m0_g0", [1, 64, 64, 64, 1], 500, 0, 0, 0.0, 3, 0.01)
    gravity_fixture(gw, "m1_g5", [1, 64, 64, 64, 1], 500, 3, 1, 5.0, 3, 0.01)


# ------------------------------------------------------------------------------------------------
def paper_fixture(tag, layers, N, seed, gamma, p):
    """Notebooks/Paper/Gross_Pitaevskii_1D_Harmonic.ipynb cell 6 (row f4): mode 0 loss = Riesz energy + PDE residual
    + 10 bc + 20 norm + 5 sym (cell 8:L103-142); -1/2 u'' + 1/2 x^2 u + gamma |u|^(p-1) u."""
    nbj = json.load(open(os.path.join(REF, "Notebooks/Paper/Gross_Pitaevskii_1D_Harmonic.ipynb")))
    from scipy.special import hermite
    ns = dict(torch=torch, nn=nn, np=np, math=math, hermite=hermite, device=torch.device("cpu"))
    exec("".join(nbj["cells"][6]["source"]), ns)
    torch.manual_seed(seed)
    lb, ub = -10.0, 10.0
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    dx = X[1, 0] - X[0, 0]
    model = ns["GrossPitaevskiiPINN"](layers, mode=0, gamma=gamma)
    flat0 = flat_params(model)
    X_tensor = torch.tensor(X, dtype=torch.float32, requires_grad=True)
    bpts = torch.tensor([[lb], [ub]], dtype=torch.float32)
    bvals = torch.zeros((2, 1), dtype=torch.float32)
    u_pred = model.forward(X_tensor)
    riesz, lam_riesz, full_u = model.riesz_loss(X_tensor, u_pred, gamma, p, "harmonic")
    pde_loss, _, lam_pde, _ = model.pde_loss(X_tensor, u_pred, gamma, p, "harmonic")
    bl = model.boundary_loss(bpts, bvals)
    nl = model.normalization_loss(model.get_complete_solution(X_tensor, u_pred), dx)
    sl = model.symmetry_loss(X_tensor, lb, ub)
    total = riesz + pde_loss + 10.0 * bl + 20.0 * nl + 5.0 * sl
    model.zero_grad()
    total.backward()
    fx = dict(layers=np.array(layers), N=N, seed=seed, mode=0, gamma=gamma, p=p, dx=dx, lb=lb, ub=ub, flat0=flat0,
              x=X.astype(np.float32), u=full_u.detach().numpy(), riesz=float(riesz), lam_pde=float(lam_pde),
              pde_loss=float(pde_loss), bc_loss=float(bl), norm_loss=float(nl), sym_loss=float(sl), total=float(total),
              grad0=flat_grads(model))
    np.savez_compressed(os.path.join(OUT, f"fx_paper_{tag}.npz"), **fx)
    print("wrote paper", tag, "riesz", fx["riesz"], "pde", fx["pde_loss"], "total", fx["total"])


if __name__ == "__main__" and os.environ.get("GOLDEN_PAPER", "1") == "1":
    paper_fixture("g10_p3", [1, 64, 64, 64, 1], 400, 0, 10.0, 3)
    paper_fixture("g2_p2", [1, 32, 32, 32, 1], 300, 1, 2.0, 2)
