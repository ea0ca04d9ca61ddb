#!/usr/bin/env python3
"""BASELINE configs[3], second half of the metric: the 2D ROTATING trap (Omega = 0.8, g = 500, complex psi, [2,128x6,2]) trained to a
vortex-lattice state, compared with the independent fp64 solver (oracle/gp_rotating_2d.py; checker only) STARTED FROM THE SAME
VORTEX-SEEDED STATE, as SURVEY 8(c) prescribes (a vortex lattice is a local minimum, one of many close in energy).

Schedule: (1) the seed -- Thomas-Fermi profile of the rotating trap times one phase winding per site of a triangular lattice of the
Feynman density Omega/pi -- is fitted by the network (gpe_mse_step, the reference's pretrain_on_analytical_solution carried to two
output channels); (2) from there the loss  |H psi - mu psi|^2 + boundary + normalisation + E[psi/|psi|] - Omega <L_z>  is minimised
with the reference's optimiser (clipped Adam), learning rate stepped down.  Two things matter (both measured):
the variational rotating-frame energy term (residual-only training drifts to excited stationary states: mu = 12.4 .. 16.3 instead of
8.73), and RE-DRAWING the collocation points (stratified, one per grid cell) every --resample epochs -- on one fixed grid the
discrete energy is driven BELOW the true minimum by structure between the points (5.78 on the training grid, 7.21 on the solver's);
(3) E, mu, <L_z>, the vortex count and |psi|^2 of the NORMALISED network state against the solver's, and against the solver
re-started from the trained network state (same basin or not).

usage: python tools/accuracy_cfg4.py [--n 256 --pretrain 30000 --epochs 120000 --train-lr 3e-4 --out profiles/r03/accuracy_cfg4_2d_6x128_rot.json]"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gpe_pinn
from gpe_pinn import capi

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256, help="training grid points per axis on [-half, half]^2")
ap.add_argument("--half", type=float, default=8.0)
ap.add_argument("--g", type=float, default=500.0)
ap.add_argument("--omega", type=float, default=0.8)
ap.add_argument("--layers", default="2,128,128,128,128,128,128,2")
ap.add_argument("--pretrain", type=int, default=30000)
ap.add_argument("--epochs", type=int, default=120000)
ap.add_argument("--lr", type=float, default=1e-3)
ap.add_argument("--w-norm", type=float, default=100.0)
ap.add_argument("--w-bc", type=float, default=10.0)
ap.add_argument("--w-riesz", type=float, default=1.0, help="weight of the variational rotating-frame energy E[psi/|psi|] - Omega <L_z>; 0: residual loss only")
ap.add_argument("--w-pde", type=float, default=1.0)
ap.add_argument("--train-lr", type=float, default=3e-4, help="learning rate of the training phase (0: --lr)")
ap.add_argument("--resample", type=int, default=100, help="epochs between changes of the collocation set (0: the fixed grid throughout)")
ap.add_argument("--sets", type=int, default=16, help="number of jittered collocation sets cycled through")
ap.add_argument("--no-basin", action="store_true", help="skip the second solver run (started from the trained network state)")
ap.add_argument("--solver-n", type=int, default=192)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--out", default="")
a = ap.parse_args()
layers = [int(v) for v in a.layers.split(",")]
g, Om, half, n = a.g, a.omega, a.half, a.n

# ---- the problem's seed (problem set-up, not checker code): TF profile x triangular lattice of phase windings ----------------------
mu_tf = math.sqrt(g * (1.0 - Om ** 2) / math.pi)
R_tf = math.sqrt(2.0 * mu_tf / (1.0 - Om ** 2))
b_lat = math.sqrt(2.0 * math.pi / (math.sqrt(3.0) * Om))
sites = []
m_ = int(R_tf / b_lat) + 2
for i in range(-m_, m_ + 1):
    for j in range(-m_, m_ + 1):
        sx, sy = b_lat * (i + 0.5 * j), b_lat * (math.sqrt(3.0) / 2.0) * j
        if math.hypot(sx, sy) <= 0.92 * R_tf:
            sites.append((sx, sy))
sites = np.array(sites)
CORE = 0.35


def seed_at(pts):
    x, y = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    r2 = x * x + y * y
    rho = np.maximum(mu_tf - 0.5 * (1.0 - Om ** 2) * r2, 0.0) / g
    psi = np.sqrt(rho + 1e-4 * np.exp(-r2 / (2.0 * (0.5 * R_tf) ** 2))).astype(np.complex128)
    z = x + 1j * y
    for (sx, sy) in sites:
        w = z - (sx + 1j * sy)
        psi = psi * w / np.sqrt(np.abs(w) ** 2 + CORE ** 2)
    return psi


ax = np.linspace(-half, half, n)
h = ax[1] - ax[0]
X = np.stack([m.ravel() for m in np.meshgrid(ax, ax, indexing="ij")], axis=1).astype(np.float32)
dv = float(h * h)
t_ = np.linspace(-half, half, 64, endpoint=False)
xb = np.concatenate([np.stack([t_, np.full_like(t_, -half)], 1), np.stack([np.full_like(t_, half), t_], 1),
                     np.stack([-t_, np.full_like(t_, half)], 1), np.stack([np.full_like(t_, -half), -t_], 1)]).astype(np.float32)
psi0 = seed_at(X)
psi0 /= math.sqrt(dv * float((np.abs(psi0) ** 2).sum()))
target = np.stack([psi0.real, psi0.imag], axis=1).astype(np.float32)

torch.manual_seed(a.seed)
import bench
flat = bench.reference_init(layers, seed=a.seed)
cfg = gpe_pinn.GPEConfig(layers=layers, gamma=g, p=3, kinetic_coeff=0.5, pot_scale=0.5, dx=dv, w_bc=a.w_bc, w_norm=a.w_norm, lr=a.lr,
                         complex_psi=True, omega_rot=Om, sched=capi.SCHED_CONST, history_capacity=8, w_pde=a.w_pde,
                         w_riesz=a.w_riesz, riesz_kind=capi.RIESZ_VARIATIONAL)
eng = gpe_pinn.Engine(cfg)
eng.set_params(flat)
xd = torch.as_tensor(X, device="cuda")
eng.bind_points(xd)
eng.bind_boundary(torch.as_tensor(xb, device="cuda"))
t0 = time.time()


def state_numbers():
    """E, mu, Lz, norm of the network state on the training grid from its output jets (fp64 sums on the host)."""
    J = eng.forward_jets(xd).cpu().numpy().astype(np.float64)            # [5][N][2]: psi, d_x, d_y, d_xx, d_yy
    pr, pi_ = J[0, :, 0], J[0, :, 1]
    x, y = X[:, 0].astype(np.float64), X[:, 1].astype(np.float64)
    rho = pr * pr + pi_ * pi_
    I = dv * rho.sum()
    kin = 0.5 * dv * (J[1] ** 2 + J[2] ** 2).sum()
    pot = dv * (0.5 * (x * x + y * y) * rho).sum()
    inter = 0.5 * g * dv * (rho * rho).sum()
    lz = dv * (pr * (x * J[2, :, 1] - y * J[1, :, 1]) - pi_ * (x * J[2, :, 0] - y * J[1, :, 0])).sum()
    En = (kin + pot) / I + inter / I ** 2 - Om * lz / I                  # energy of the normalised state
    mun = (kin + pot) / I + 2.0 * inter / I ** 2 - Om * lz / I
    return dict(E=float(En), mu=float(mun), lz=float(lz / I), norm=float(I), psi=(pr + 1j * pi_) / math.sqrt(I))


def windings(psi2d, hgrid):
    ph = np.angle(psi2d)
    dw = lambda v: (v + np.pi) % (2.0 * np.pi) - np.pi
    c = dw(ph[1:, :-1] - ph[:-1, :-1]) + dw(ph[1:, 1:] - ph[1:, :-1]) + dw(ph[:-1, 1:] - ph[1:, 1:]) + dw(ph[:-1, :-1] - ph[:-1, 1:])
    w = np.rint(c / (2.0 * np.pi)).astype(int)
    rho = np.abs(psi2d) ** 2
    from scipy.ndimage import gaussian_filter
    sm = gaussian_filter(rho, 0.6 / hgrid)
    mask = 0.25 * (sm[1:, :-1] + sm[:-1, :-1] + sm[1:, 1:] + sm[:-1, 1:]) > 0.02 * sm.max()
    return int(((w > 0) & mask).sum()), int(((w < 0) & mask).sum())


# ---- (1) fit the seed ----------------------------------------------------------------------------------------------------------------
eng.bind_target(torch.as_tensor(target, device="cuda"))
eng.reset_optimizer(a.lr)
for i in range(a.pretrain):
    if i == int(0.6 * a.pretrain):
        eng.set_lr(a.lr * 0.3)
    if i == int(0.85 * a.pretrain):
        eng.set_lr(a.lr * 0.1)
    if i % 5000 == 0 or i == a.pretrain - 1:
        sc = eng.mse_step()
        print(f"pretrain {i}: mse {sc['loss']:.3e} ({time.time() - t0:.0f} s)", flush=True)
    else:
        eng.lib.gpe_mse_begin(eng._h); eng.lib.gpe_mse_update(eng._h)
pre_mse = sc["loss"]
eng.bind_target(None)
s0 = state_numbers()
v0 = windings(s0["psi"].reshape(n, n), h)
print(f"after the fit: E {s0['E']:.5f} mu {s0['mu']:.5f} Lz {s0['lz']:.4f} norm {s0['norm']:.5f} vortices {v0}", flush=True)

# ---- (2) residual training from the seeded network ---------------------------------------------------------------------------------------
tlr = a.train_lr or a.lr
eng.reset_optimizer(tlr)
rows = []
# Collocation sets: the energy and the residual are SUMS over the collocation points; on one fixed grid a long enough training run
# lowers the discrete energy below the true minimum by growing structure between the points (measured: E = 5.78 on the 256^2 training
# grid, 7.21 for the same network on the solver's grid with spectral derivatives).  The points are therefore re-drawn every
# --resample epochs: stratified sampling, one uniformly placed point per grid cell (the quadrature weight stays the cell area).
rng = np.random.default_rng(a.seed + 1)
cells = np.stack([m.ravel() for m in np.meshgrid(ax, ax, indexing="ij")], axis=1)
sets = [xd]
for _ in range(a.sets - 1 if a.resample > 0 else 0):
    sets.append(torch.as_tensor((cells + rng.uniform(-0.5 * h, 0.5 * h, cells.shape)).astype(np.float32), device="cuda"))
epoch_ctr = 0
for frac, lr in ((0.35, tlr), (0.2, tlr * 0.3), (0.15, tlr * 0.1), (0.1, tlr * 0.03), (0.1, tlr * 0.01), (0.05, tlr * 0.003), (0.05, tlr * 0.001)):
    eng.set_lr(lr)
    left = int(a.epochs * frac)
    while left > 0:
        k = min(left, 10000)
        if a.resample > 0:
            done = 0
            while done < k:
                eng.bind_points(sets[(epoch_ctr // a.resample) % len(sets)])
                kk = min(a.resample - epoch_ctr % a.resample, k - done)
                eng.run(kk)
                done += kk
                epoch_ctr += kk
        else:
            eng.run(k)
        left -= k
        sc = eng.read_scalars()
        print(f"   lr {lr:.1e} loss {sc['loss']:.3e} pde {sc['pde']:.3e} mu {sc['mu']:.6f} E {sc['riesz']:.6f} int {sc['integral']:.6f} ({time.time() - t0:.0f} s)", flush=True)
    rows.append(dict(lr=lr, epochs=int(a.epochs * frac), loss=sc["loss"], pde=sc["pde"], mu=sc["mu"], norm=sc["integral"], energy=sc["riesz"]))
wall = time.time() - t0
eng.bind_points(xd)
s1 = state_numbers()
v1 = windings(s1["psi"].reshape(n, n), h)
print(f"trained: E {s1['E']:.6f} mu {s1['mu']:.6f} (Rayleigh quotient of the raw state {sc['mu']:.6f}) Lz {s1['lz']:.4f} norm {s1['norm']:.6f} vortices {v1}", flush=True)

# ---- (3) the checker: oracle/gp_rotating_2d.py from the same seed -------------------------------------------------------------------------
from oracle import gp_rotating_2d as R
bx = R.Box(a.solver_n, 12.0)
ts = time.time()
ref = R.minimise(bx, R.seed_state(bx, g, Om, sites, core=CORE), g, Om, tol=1e-8, max_iter=8000)
vref = R.count_vortices(bx, ref["psi"])
dref = R.interp_density(bx, ref["psi"], X.astype(np.float64))
dref /= dv * dref.sum()
dens = np.abs(s1["psi"]) ** 2
dens /= dv * dens.sum()
rel_l2 = float(np.sqrt(((dens - dref) ** 2).sum() / (dref ** 2).sum()))
# the trap is isotropic: the lattice as a whole may turn (a zero mode of the energy) -- density error also after the best rigid rotation
best = (rel_l2, 0.0)
Xd64 = X.astype(np.float64)
for th in np.linspace(-0.55, 0.55, 221):          # (a triangular lattice repeats every pi/3)
    c_, s_ = math.cos(th), math.sin(th)
    Xr = np.stack([c_ * Xd64[:, 0] - s_ * Xd64[:, 1], s_ * Xd64[:, 0] + c_ * Xd64[:, 1]], axis=1)
    dr = R.interp_density(bx, ref["psi"], np.clip(Xr, -11.9, 11.9))
    dr /= dv * dr.sum()
    e_ = float(np.sqrt(((dens - dr) ** 2).sum() / (dr ** 2).sum()))
    if e_ < best[0]:
        best = (e_, float(th))
rel_l2_rot, theta_rot = best
print(f"solver ({a.solver_n}^2, {time.time() - ts:.0f} s): E {ref['E']:.6f} mu {ref['mu']:.6f} Lz {ref['lz']:.4f} vortices {vref[0]} res {ref['residual']:.1e}")
basin = None
if not a.no_basin:
    # ... and the same solver started from the TRAINED NETWORK STATE (sampled on the solver grid, zero outside the training box): the
    # stationary state of the basin the training ended in -- clipped Adam hops between the many lattice arrangements the seed's basin
    # borders on, conjugate gradients do not, so the two runs need not end in the same arrangement
    Xs = np.stack([bx.X.ravel(), bx.Y.ravel()], axis=1)
    inside = (np.abs(Xs[:, 0]) <= half) & (np.abs(Xs[:, 1]) <= half)
    pn = np.zeros(Xs.shape[0], dtype=np.complex128)
    o = eng.forward(torch.as_tensor(Xs[inside].astype(np.float32), device="cuda")).cpu().numpy().astype(np.float64)
    pn[inside] = o[:, 0] + 1j * o[:, 1]
    ts = time.time()
    pol = R.minimise(bx, pn.reshape(bx.n, bx.n), g, Om, tol=1e-8, max_iter=8000)
    vpol = R.count_vortices(bx, pol["psi"])
    e_nn_on_solver_grid = R.energy_parts(bx, pn.reshape(bx.n, bx.n) / math.sqrt(bx.dv * float((np.abs(pn) ** 2).sum())), g, Om)
    dpol = R.interp_density(bx, pol["psi"], X.astype(np.float64))
    dpol /= dv * dpol.sum()
    rel_l2_pol = float(np.sqrt(((dens - dpol) ** 2).sum() / (dpol ** 2).sum()))
    print(f"solver from the network state ({pol['iterations']} iterations, {time.time() - ts:.0f} s): E {pol['E']:.6f} mu {pol['mu']:.6f} Lz {pol['lz']:.4f} "
          f"vortices {vpol[0]}; network state on the solver grid (spectral derivatives): E {e_nn_on_solver_grid['E']:.6f} mu {e_nn_on_solver_grid['mu']:.6f}; "
          f"density rel L2 vs this state {rel_l2_pol:.2e}")
    basin = dict(note="the solver started from the trained network state: the stationary state of the basin the training ended in",
                 E_ref=pol["E"], mu_ref=pol["mu"], lz_ref=pol["lz"], vortices_ref=vpol[0], iterations=pol["iterations"],
                 residual=pol["residual"], E_abs_err=abs(s1["E"] - pol["E"]), mu_abs_err=abs(s1["mu"] - pol["mu"]),
                 density_rel_l2=rel_l2_pol, network_state_on_solver_grid=dict(E=e_nn_on_solver_grid["E"], mu=e_nn_on_solver_grid["mu"]))
out = dict(case="cfg4_2d_rot", workload="cfg4_2d_6x128_rot", layers=layers, points=int(X.shape[0]), grid_per_axis=n, g=g, omega_rot=Om,
           seed_sites=len(sites), lattice_spacing=b_lat, pretrain_steps=a.pretrain, pretrain_mse=pre_mse, epochs=a.epochs, stages=rows,
           wall_seconds=wall, after_fit=dict(E=s0["E"], mu=s0["mu"], lz=s0["lz"], vortices=v0[0], antivortices=v0[1]),
           E=s1["E"], mu=s1["mu"], mu_rayleigh_raw=sc["mu"], lz=s1["lz"], norm_integral=s1["norm"], vortices=v1[0], antivortices=v1[1],
           E_ref=ref["E"], mu_ref=ref["mu"], lz_ref=ref["lz"], vortices_ref=vref[0], solver_residual=ref["residual"],
           solver=f"oracle/gp_rotating_2d.py, {a.solver_n}^2 Fourier grid on [-12,12)^2, same seed",
           E_abs_err=abs(s1["E"] - ref["E"]), mu_abs_err=abs(s1["mu"] - ref["mu"]), density_rel_l2=rel_l2, final_pde_loss=sc["pde"],
           basin=basin, density_rel_l2_best_rotation=rel_l2_rot, rotation_angle=theta_rot,
           schedule=dict(lr=a.lr, train_lr=tlr, resample_every=a.resample, collocation_sets=len(sets), w_norm=a.w_norm, w_bc=a.w_bc, w_pde=a.w_pde, w_riesz=a.w_riesz, ladder="(0.35, 0.2, 0.15, 0.1, 0.1, 0.05, 0.05) of the epochs at lr x (1, .3, .1, .03, .01, .003, .001)"))
path = a.out or os.path.join(ROOT, "gpurun_out", "accuracy_cfg4_2d_6x128_rot.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
if basin:
    print(f"basin: E {s1['E']:.6f} vs {basin['E_ref']:.6f} (|err| {basin['E_abs_err']:.2e})  mu {s1['mu']:.6f} vs {basin['mu_ref']:.6f} (|err| {basin['mu_abs_err']:.2e})  "
          f"vortices {v1[0]} vs {basin['vortices_ref']}  density rel L2 {basin['density_rel_l2']:.2e}")
print(f"density rel L2 after the best rigid rotation ({theta_rot:+.4f} rad): {rel_l2_rot:.2e}")
print(f"E {s1['E']:.6f} vs {ref['E']:.6f} (|err| {out['E_abs_err']:.2e})  mu {s1['mu']:.6f} vs {ref['mu']:.6f} (|err| {out['mu_abs_err']:.2e})  "
      f"vortices {v1[0]} vs {vref[0]}  density rel L2 {rel_l2:.2e}  {wall:.0f} s")
