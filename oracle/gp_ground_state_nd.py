"""
oracle/gp_ground_state_nd.py -- independent fp64 ground truth for the Gross-Pitaevskii ground state in 1, 2 and 3 dimensions.
TEST INFRASTRUCTURE ONLY (checker for the |mu - mu_ref| half of the BASELINE metric; never imported by the product).

Solves the stationary equation of BASELINE.json's north star on a periodic box,

    -1/2 lap(u) + 1/2 sum_k (omega_k x_k)^2 u + g u^3 = mu u ,      h^d sum u^2 = 1 ,

with a Fourier-spectral Laplacian (the ground state decays like a Gaussian, so the periodic images are below 1e-30) and
Newton's method on the bordered system  [[J, -u], [-u^T, 0]] [du; dmu] = [-F; G/(2 h^d)],  J = -1/2 lap + V + 3 g u^2 - mu,
solved matrix-free with MINRES preconditioned by (-1/2 lap + sigma)^-1 (FFT-diagonal).  Continuation in g from the analytic
g = 0 Gaussian.  There is no time step, hence no splitting error: the only discretisation parameter is the grid, and the result
is checked for grid independence (two resolutions).  This is NOT reference code -- the reference has no 2D/3D harmonic problem
and its own FDM notebook has the wrong Laplacian sign.

Known answers reproduced (tests/test_ground_state_cpu.py): g = 0 -> mu = 1/2 sum omega_k exactly; 1D g = 100 agrees with the
finite-difference Newton solver of oracle/gp_ground_state.py; large g -> Thomas-Fermi value; SURVEY 8(c)'s scratch values
12.678 (2D, g = 500) and 13.089 (3D, omega = (1, 1.4, 2), g = 1000) to their stated ~1e-3.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse.linalg as spla


class Grid:
    def __init__(self, n, half):
        self.n = tuple(int(v) for v in n)
        self.half = tuple(float(v) for v in half)
        self.d = len(self.n)
        self.axes = tuple(range(self.d))
        self.h = [2.0 * a / m for a, m in zip(self.half, self.n)]
        self.x = [-a + hk * np.arange(m) for a, hk, m in zip(self.half, self.h, self.n)]
        self.dv = float(np.prod(self.h))
        ks = [2.0 * np.pi * np.fft.fftfreq(m, d=hk) for m, hk in zip(self.n, self.h)]
        ks[-1] = 2.0 * np.pi * np.fft.rfftfreq(self.n[-1], d=self.h[-1])
        K = np.meshgrid(*ks, indexing="ij")
        self.k2 = sum(k * k for k in K)

    def mesh(self):
        return np.meshgrid(*self.x, indexing="ij")

    def lap(self, u):
        return np.fft.irfftn(-self.k2 * np.fft.rfftn(u), s=self.n, axes=self.axes)

    def inv_kin(self, r, sigma):
        """(-1/2 lap + sigma)^-1 r"""
        return np.fft.irfftn(np.fft.rfftn(r) / (0.5 * self.k2 + sigma), s=self.n, axes=self.axes)


def _newton(grid: Grid, V, g, u, mu, tol=1e-11, verbose=False):
    n = u.size
    shape = u.shape
    dv = grid.dv
    for it in range(40):
        Hu = -0.5 * grid.lap(u) + (V + g * u * u) * u
        F = Hu - mu * u
        G = dv * float((u * u).sum()) - 1.0
        rn = math.sqrt(dv * float((F * F).sum()))
        if verbose:
            print(f"      newton {it}: |F| {rn:.3e} G {G:.3e} mu {mu:.10f}")
        if rn < tol and abs(G) < tol:
            break
        W = V + 3.0 * g * u * u - mu
        sigma = max(float(W.mean()), 0.5)
        su = math.sqrt(dv)                      # scale the border so that the bordered operator is symmetric in the dv inner product

        def matvec(z):
            du = z[:n].reshape(shape)
            dm = z[n]
            top = -0.5 * grid.lap(du) + W * du - dm * su * u
            bot = -su * float((u * du).sum())
            return np.concatenate([top.ravel(), [bot]])

        def prec(z):
            r = z[:n].reshape(shape)
            return np.concatenate([grid.inv_kin(r, sigma).ravel(), [z[n]]])

        A = spla.LinearOperator((n + 1, n + 1), matvec=matvec, dtype=np.float64)
        M = spla.LinearOperator((n + 1, n + 1), matvec=prec, dtype=np.float64)
        rhs = np.concatenate([-F.ravel(), [G / (2.0 * dv) * su]])
        z, info = spla.minres(A, rhs, M=M, rtol=1e-10, maxiter=2000)
        u = u + z[:n].reshape(shape)
        mu = mu + z[n] * su
    return u, mu, rn


def ground_state(omega, g, n, half, g_steps=None, verbose=False):
    """-> dict(mu, energy, u, grid, residual).  omega: trap frequencies (length = dimension); g: interaction strength."""
    omega = [float(w) for w in omega]
    grid = Grid(n, half)
    X = grid.mesh()
    V = 0.5 * sum((w * x) ** 2 for w, x in zip(omega, X))
    u = np.ones(grid.n)
    for w, x in zip(omega, X):
        u = u * (w / math.pi) ** 0.25 * np.exp(-0.5 * w * x * x)          # exact g = 0 ground state
    mu = 0.5 * sum(omega)
    if g_steps is None:
        g_steps = [s for s in (1, 4, 12, 30, 60, 100, 160, 240, 350, 500, 700, 1000, 1500, 2000, 3000, 5000) if s < g] + [g]
    res = 0.0
    for gi in ([0.0] if g == 0 else g_steps):
        if verbose:
            print(f"   g = {gi}")
        u, mu, res = _newton(grid, V, float(gi), u, mu, verbose=verbose)
    kin = -0.5 * grid.dv * float((u * grid.lap(u)).sum())
    pot = grid.dv * float((V * u * u).sum())
    inter = 0.5 * g * grid.dv * float((u ** 4).sum())
    return dict(mu=float(mu), energy=kin + pot + inter, u=u, grid=grid, residual=res,
                mu_from_energy=kin + pot + 2.0 * inter)


def density_on(grid: Grid, u, pts):
    """|u|^2 at arbitrary points by trigonometric interpolation (exact for the band-limited solution)."""
    from scipy.interpolate import RegularGridInterpolator
    axes = [np.append(x, -x[0]) for x in grid.x]                                   # close the periodic box
    up = u
    for ax in range(grid.d):
        up = np.concatenate([up, np.take(up, [0], axis=ax)], axis=ax)
    f = RegularGridInterpolator(axes, up, method="cubic" if grid.d <= 2 else "linear")
    return f(pts) ** 2


def thomas_fermi_mu(omega, g):
    d = len(omega)
    wbar = float(np.prod(omega)) ** (1.0 / d)
    if d == 1:
        return (3.0 * g * wbar / (4.0 * math.sqrt(2.0))) ** (2.0 / 3.0)
    if d == 2:
        return math.sqrt(g / math.pi) * wbar
    return 0.5 * (15.0 * g * wbar ** 3 / (4.0 * math.pi)) ** (2.0 / 5.0)


if __name__ == "__main__":
    import json
    import os
    import sys
    import time
    out = {}
    cases = {
        "1d_g100": (dict(omega=[1.0], g=100.0), [dict(n=[512], half=[16.0]), dict(n=[768], half=[16.0])]),
        "2d_g500": (dict(omega=[1.0, 1.0], g=500.0), [dict(n=[160, 160], half=[10.0, 10.0]), dict(n=[224, 224], half=[10.0, 10.0])]),
        "3d_aniso_g1000": (dict(omega=[1.0, 1.4, 2.0], g=1000.0),
                           [dict(n=[80, 64, 48], half=[8.0, 6.0, 4.5]), dict(n=[112, 88, 64], half=[8.0, 6.0, 4.5])]),
    }
    only = sys.argv[1:] or list(cases)
    for name in only:
        pb, grids = cases[name]
        rows = []
        for gr in grids:
            t0 = time.time()
            r = ground_state(pb["omega"], pb["g"], gr["n"], gr["half"], verbose=False)
            rows.append(dict(n=gr["n"], half=gr["half"], mu=r["mu"], energy=r["energy"], residual=r["residual"],
                             mu_from_energy=r["mu_from_energy"], seconds=time.time() - t0))
            print(name, rows[-1], flush=True)
        out[name] = dict(problem=pb, grids=rows, mu=rows[-1]["mu"], energy=rows[-1]["energy"],
                         grid_independence=abs(rows[-1]["mu"] - rows[0]["mu"]), thomas_fermi_mu=thomas_fermi_mu(pb["omega"], pb["g"]))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gp_ground_truth.json")
    prev = json.load(open(path)) if os.path.exists(path) else {}
    prev.update(out)
    json.dump(prev, open(path, "w"), indent=1)
    print("wrote", path)
