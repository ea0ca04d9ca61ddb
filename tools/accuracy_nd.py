#!/usr/bin/env python3
"""BASELINE metric, second half, for the d > 1 / large-g configurations: train the engine to the ground state and compare the
Rayleigh quotient mu and the density |psi|^2 on a test grid with the independent fp64 spectral-Newton solver
(oracle/gp_ground_state_nd.py -> oracle/gp_ground_truth.json; checker only).

Schedule (the reference's own strategy, refine/harmonic_pinn_simulation.py:286-407, carried to d dimensions): pre-train the
network on the analytic g = 0 ground state (pretrain_on_analytical_solution :650-701 -> gpe_mse_step), then continue in gamma
(gpe_set_gamma) with a fresh Adam + ReduceLROnPlateau per stage, warm-starting from the previous stage's weights.

usage: python tools/accuracy_nd.py --case ns_2d [--epochs 4000 --final 60000 --n 256]"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gpe_pinn
from gpe_pinn import capi

CASES = {
    "ns_2d": dict(layers=[2, 64, 64, 64, 64, 1], omega=(1.0, 1.0), g=500.0, half=8.0, truth="2d_g500", workload="ns_2d_4x64"),
    "cfg2_1d": dict(layers=[1, 64, 64, 64, 64, 1], omega=(1.0,), g=100.0, half=10.0, truth="1d_g100", workload="cfg2_1d_4x64"),
    "cfg3_2d": dict(layers=[2, 128, 128, 128, 128, 128, 1], omega=(1.0, 1.0), g=500.0, half=8.0, truth="2d_g500", workload="cfg3_2d_5x128"),
    "cfg5_3d": dict(layers=[3, 256, 256, 256, 256, 256, 256, 1], omega=(1.0, 1.4, 2.0), g=1000.0, half=6.0, truth="3d_aniso_g1000",
                    workload="cfg5_3d_6x256"),
}
ap = argparse.ArgumentParser()
ap.add_argument("--case", default="ns_2d", choices=sorted(CASES))
ap.add_argument("--n", type=int, default=0, help="grid points per axis (0: 256 in 2D, 65536 in 1D, 64 in 3D)")
ap.add_argument("--pretrain", type=int, default=3000)
ap.add_argument("--epochs", type=int, default=4000, help="epochs per continuation stage")
ap.add_argument("--final", type=int, default=60000, help="epochs of the last stage")
ap.add_argument("--stages", type=int, default=16)
ap.add_argument("--lr", type=float, default=1e-3)
ap.add_argument("--w-norm", type=float, default=20.0)
ap.add_argument("--w-norm-final", type=float, default=100.0, help="normalisation weight of the last stage (mu scales with the norm: g |u|^2)")
ap.add_argument("--w-bc", type=float, default=10.0)
ap.add_argument("--w-riesz", type=float, default=1.0, help="weight of the variational (normalised-state) energy term; 0: residual loss only")
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--resample", type=int, default=0, help="epochs between changes of the collocation set: stratified sampling, ONE uniformly placed point per grid "
                                                       "cell, quadrature weight = the cell volume (0: the fixed grid throughout).  On one fixed grid a long run fits the "
                                                       "residual AT the points and grows structure between them (cfg5 at 40^3: mu 13.018 on the grid, 26.2 on a finer one)")
ap.add_argument("--sets", type=int, default=16, help="number of jittered collocation sets cycled through")
ap.add_argument("--big-grid", default="", help="e.g. 64,128,64: after the schedule, continue on THIS grid (BASELINE's per-GPU size) for --big-epochs epochs "
                                               "at the low end of the learning-rate ladder, and evaluate mu there (same weights, fresh Adam)")
ap.add_argument("--big-epochs", type=int, default=3000)
ap.add_argument("--out", default="")
a = ap.parse_args()
cs = CASES[a.case]
d = len(cs["omega"])
n = a.n or {1: 65536, 2: 256, 3: 64}[d]
half = cs["half"]
axes = [np.linspace(-half, half, n) for _ in range(d)]
h = axes[0][1] - axes[0][0]
X = np.stack([m.ravel() for m in np.meshgrid(*axes, indexing="ij")], axis=1).astype(np.float32)
dv = float(h ** d)
# boundary points: the faces of the box (1D: the two ends)
if d == 1:
    xb = np.array([[-half], [half]], np.float32)
else:
    t = np.linspace(-half, half, 64 if d == 2 else 12, endpoint=False)
    faces = []
    for ax in range(d):
        others = np.meshgrid(*([t] * (d - 1)), indexing="ij")
        for side in (-half, half):
            cols = [o.ravel() for o in others]
            cols.insert(ax, np.full(cols[0].shape, side))
            faces.append(np.stack(cols, axis=1))
    xb = np.concatenate(faces).astype(np.float32)

truth = json.load(open(os.path.join(ROOT, "oracle", "gp_ground_truth.json")))[cs["truth"]]
mu_ref = truth["mu"]

torch.manual_seed(a.seed)
import bench
flat = bench.reference_init(cs["layers"], seed=a.seed)
cfg = gpe_pinn.GPEConfig(layers=cs["layers"], gamma=0.0, p=3, kinetic_coeff=0.5, pot_scale=0.5, omega=tuple(cs["omega"]) + (1.0,) * (3 - d),
                         dx=dv, w_bc=a.w_bc, w_norm=a.w_norm, lr=a.lr, sched=capi.SCHED_CONST, history_capacity=8,
                         w_riesz=a.w_riesz, riesz_kind=capi.RIESZ_VARIATIONAL)
eng = gpe_pinn.Engine(cfg)
eng.set_params(flat)
xd = torch.as_tensor(X, device="cuda")
eng.bind_points(xd)
eng.bind_boundary(torch.as_tensor(xb, device="cuda"))
# stratified collocation sets (--resample): the grid point moved uniformly inside its cell; the first and last cells of an axis stay inside the box
xsets = [xd]
if a.resample > 0:
    rng = np.random.default_rng(1234 + a.seed)
    for _ in range(a.sets):
        J = X.astype(np.float64) + rng.uniform(-0.5 * h, 0.5 * h, X.shape)
        xsets.append(torch.as_tensor(np.clip(J, -half, half).astype(np.float32), device="cuda"))
_set_i = [0]


def run_epochs(n_ep):
    """eng.run(n_ep), changing the collocation set every --resample epochs (bind_points is a pointer swap)"""
    if a.resample <= 0:
        eng.run(n_ep)
        return
    left = n_ep
    while left > 0:
        k = min(left, a.resample)
        _set_i[0] = (_set_i[0] + 1) % len(xsets)
        eng.bind_points(xsets[_set_i[0]])
        eng.run(k)
        left -= k


t0 = time.time()
# ---- pre-training on the analytic g = 0 ground state (product of Gaussians) ----
phi0 = np.ones(X.shape[0])
for k, w in enumerate(cs["omega"]):
    phi0 = phi0 * (w / math.pi) ** 0.25 * np.exp(-0.5 * w * X[:, k].astype(np.float64) ** 2)
eng.bind_target(torch.as_tensor(phi0.astype(np.float32), device="cuda"))
eng.reset_optimizer(a.lr)
for i in range(a.pretrain):
    sc = eng.mse_step() if (i % 500 == 0 or i == a.pretrain - 1) else None
    if sc is None:
        eng.lib.gpe_mse_begin(eng._h); eng.lib.gpe_mse_update(eng._h)
    elif i % 500 == 0:
        print(f"pretrain {i}: mse {sc['loss']:.3e}", flush=True)
eng.bind_target(None)
# ---- gamma continuation ----
gam = [cs["g"] * (k / a.stages) ** 2 for k in range(a.stages + 1)]            # quadratic ramp: small steps where the state changes fastest
rows = []
for si, g in enumerate(gam):
    eng.set_gamma(g)
    eng.reset_optimizer(a.lr)
    last = si == len(gam) - 1
    ne = a.final if last else a.epochs
    if not last:
        run_epochs(ne)
    else:                                   # last stage: tighter normalisation, learning rate stepped down (Adam state kept)
        eng.set_loss_weights(1.0, a.w_bc, a.w_norm_final, 0.0, 0.0, a.w_riesz)
        # (the ladder ends at lr x 1e-3: Adam turns gradient round-off into steps of size lr, and mu follows the norm with
        #  d mu / d int ~ 8 -- at lr 1e-5 the norm integral wanders by ~2e-4, i.e. mu by ~1.5e-3 for some seeds)
        for frac, lr in ((0.35, a.lr), (0.2, a.lr * 0.3), (0.15, a.lr * 0.1), (0.1, a.lr * 0.03), (0.1, a.lr * 0.01), (0.05, a.lr * 0.003),
                         (0.05, a.lr * 0.001)):
            eng.set_lr(lr)
            left = int(ne * frac)
            while left > 0:                       # progress line at least every ~10 000 epochs (a silent GPU job is taken to be hung)
                n_run = min(left, 10000)
                run_epochs(n_run)
                left -= n_run
                scp = eng.read_scalars()
                print(f"   final stage: lr {lr:.1e} mu {scp['mu']:.6f} pde {scp['pde']:.3e} int {scp['integral']:.6f} ({time.time() - t0:.0f} s)", flush=True)
    sc = eng.read_scalars()
    if last and a.resample > 0:                 # the reported numbers: on the REGULAR grid, not on the last jittered set
        eng.bind_points(xd)
        sc = eng.residual(want_fields=False)[0]
    rows.append(dict(gamma=g, epochs=ne, mu=sc["mu"], loss=sc["loss"], pde=sc["pde"], norm=sc["integral"], lr=sc["lr"], riesz=sc["riesz"]))
    print(f"stage {si}: gamma {g:8.2f} mu {sc['mu']:.6f} E {sc['riesz']:.6f} loss {sc['loss']:.3e} pde {sc['pde']:.3e} int {sc['integral']:.6f} lr {sc['lr']:.1e} "
          f"({time.time() - t0:.0f} s)", flush=True)
big = None
if a.big_grid:
    # the per-GPU batch of the BASELINE configuration: same box, finer grid; the trained weights move over, the quadrature weight changes
    nb = [int(v) for v in a.big_grid.split(",")]
    assert len(nb) == d
    axb = [np.linspace(-half, half, k) for k in nb]
    Xb = np.stack([m.ravel() for m in np.meshgrid(*axb, indexing="ij")], axis=1).astype(np.float32)
    dvb = float(np.prod([ax[1] - ax[0] for ax in axb]))
    cfg_b = gpe_pinn.GPEConfig(layers=cs["layers"], gamma=cs["g"], p=3, kinetic_coeff=0.5, pot_scale=0.5, omega=tuple(cs["omega"]) + (1.0,) * (3 - d),
                               dx=dvb, w_bc=a.w_bc, w_norm=a.w_norm_final, lr=a.lr * 0.01, sched=capi.SCHED_CONST, history_capacity=8,
                               w_riesz=a.w_riesz, riesz_kind=capi.RIESZ_VARIATIONAL)
    eng_b = gpe_pinn.Engine(cfg_b)
    eng_b.set_params(eng.get_params())
    mu_small = rows[-1]["mu"]
    eng.close()
    eng = eng_b
    xdb = torch.as_tensor(Xb, device="cuda")
    eng.bind_points(xdb)
    eng.bind_boundary(torch.as_tensor(xb, device="cuda"))
    sc0 = eng.residual(want_fields=False)[0]
    print(f"big grid {nb} ({Xb.shape[0]} points): mu of the small-grid state {sc0['mu']:.6f} (on its own grid {mu_small:.6f})", flush=True)
    tb = time.time()
    for frac, lr in (((0.5, a.lr * 0.01), (0.3, a.lr * 0.003), (0.2, a.lr * 0.001)) if a.big_epochs > 0 else ()):
        eng.set_lr(lr)
        left = int(a.big_epochs * frac)
        while left > 0:
            n_run = min(left, 500)
            eng.run(n_run)
            left -= n_run
            scp = eng.read_scalars()
            print(f"   big grid: lr {lr:.1e} mu {scp['mu']:.6f} pde {scp['pde']:.3e} int {scp['integral']:.6f} ({time.time() - t0:.0f} s)", flush=True)
    sc = eng.read_scalars() if a.big_epochs > 0 else sc0
    big = dict(grid=nb, points=int(Xb.shape[0]), epochs=a.big_epochs, mu_before=sc0["mu"], seconds=time.time() - tb)
    rows.append(dict(gamma=cs["g"], epochs=a.big_epochs, mu=sc["mu"], loss=sc["loss"], pde=sc["pde"], norm=sc["integral"], lr=sc["lr"], riesz=sc["riesz"]))
wall = time.time() - t0
mu = rows[-1]["mu"]
# mu of the NORMALISED state u / sqrt(int), from the three scalars of the last stage: with A = (kinetic + potential) / int and
# B = g int(u^4) / int, the Rayleigh quotient is A + B and the variational energy of the normalised state A + B / (2 int); the
# chemical potential of the normalised state is A + B / int.  (d mu / d int = B ~ 8 here: a norm that is off by 1e-4 moves the
# raw quotient by 1e-3 although the state itself -- energy, density -- is right.)
_int, _E = rows[-1]["norm"], rows[-1]["riesz"]
mu_normalised = None
if a.w_riesz != 0.0 and _int > 0.5:
    _B = (mu - _E) / (1.0 - 0.5 / _int)
    mu_normalised = (mu - _B) + _B / _int
# ---- density on a test grid vs the solver ----
from oracle import gp_ground_state_nd as nd
gr = truth["grids"][0]
sol = nd.ground_state(truth["problem"]["omega"], truth["problem"]["g"], gr["n"], gr["half"])
nt = {1: 1000, 2: 96, 3: 24}[d]
tax = [np.linspace(-min(0.9 * half, 0.95 * gr["half"][k]), min(0.9 * half, 0.95 * gr["half"][k]), nt) for k in range(d)]
XT = np.stack([m.ravel() for m in np.meshgrid(*tax, indexing="ij")], axis=1)
ht = float(np.prod([t[1] - t[0] for t in tax]))
u, dens = eng.eval_density(torch.as_tensor(XT.astype(np.float32), device="cuda"), ht)
dens = dens.cpu().numpy().astype(np.float64)
dens = dens / (dens.sum() * ht)                             # both normalised on the test grid
dref = nd.density_on(sol["grid"], sol["u"], XT)
dref = dref / (dref.sum() * ht)
out = dict(case=a.case, workload=cs["workload"], layers=cs["layers"], points=int(X.shape[0]), grid_per_axis=n, stages=rows,
           total_epochs=int(sum(r["epochs"] for r in rows)) + a.pretrain, wall_seconds=wall,
           mu=mu, mu_ref=mu_ref, mu_abs_err=abs(mu - mu_ref), mu_ref_source="oracle/gp_ground_truth.json:" + cs["truth"],
           mu_normalised_state=mu_normalised, mu_normalised_state_abs_err=(abs(mu_normalised - mu_ref) if mu_normalised is not None else None),
           norm_integral=_int, seed=a.seed,
           density_max_abs_err=float(np.abs(dens - dref).max()), density_max=float(dref.max()),
           density_rel_l2=float(np.sqrt(((dens - dref) ** 2).sum() / (dref ** 2).sum())),
           schedule=dict(pretrain=a.pretrain, epochs=a.epochs, final=a.final, stages=a.stages, lr=a.lr, w_norm=a.w_norm, w_bc=a.w_bc,
                         w_riesz=a.w_riesz, w_norm_final=a.w_norm_final, resample=a.resample, sets=a.sets,
                         scheduler="constant lr per stage, fresh Adam per stage; last stage lr x (1, 0.3, 0.1, 0.03, 0.01, 0.003, 0.001)"),
           energy=rows[-1]["riesz"], energy_ref=truth["energy"], big_grid=big)
path = a.out or os.path.join(ROOT, "gpurun_out", f"accuracy_{cs['workload']}.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
print(f"mu {mu:.6f}  mu_ref {mu_ref:.6f}  |err| {abs(mu - mu_ref):.2e}   normalised state: mu {mu_normalised if mu_normalised is not None else float('nan'):.6f} "
      f"|err| {abs(mu_normalised - mu_ref) if mu_normalised is not None else float('nan'):.2e} (int {_int:.6f})   density rel L2 {out['density_rel_l2']:.2e}   {wall:.0f} s")
