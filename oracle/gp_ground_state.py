"""
oracle/gp_ground_state.py -- independent fp64 ground truth for the 1D Gross-Pitaevskii eigenvalue.  TEST INFRASTRUCTURE ONLY.

Solves   -c u'' + V(x) u + gamma u^3 = lam u ,  dx * sum u^2 = 1   (refine convention: c = 1, V = x^2,
refine/harmonic_pinn_simulation.py:181-188; notebook convention: c = 1/2, V = x^2/2, nb c6:L113-121)
by Newton's method on a bordered second-order finite-difference system, continued in gamma, with Richardson
extrapolation h -> h/2.  This is NOT reference code (the reference's own FDM notebook has the wrong Laplacian sign,
SURVEY 2.1); it is the yardstick for the |mu - mu_ref| half of the BASELINE metric where no stored reference value exists.
Known answers it reproduces: gamma = 0 -> lam = 2n+1 (c=1) / n+1/2 (c=1/2); Thomas-Fermi limit for large gamma.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def _solve_on_grid(gammas, n, half, c, vscale, mode):
    x = np.linspace(-half, half, n)
    h = x[1] - x[0]
    V = vscale * x * x
    lap = sp.diags([np.full(n - 1, 1.0), np.full(n, -2.0), np.full(n - 1, 1.0)], [-1, 0, 1]) / (h * h)
    A0 = (-c * lap + sp.diags(V)).tocsc()
    # linear start: eigenvector number `mode` of the linear operator
    w, v = spla.eigsh(A0, k=mode + 1, sigma=0.0, which="LM")
    order = np.argsort(w)
    u = v[:, order[mode]]
    u = u / np.sqrt(h * (u * u).sum())
    if u[np.argmax(np.abs(u))] < 0:
        u = -u
    lam = float(w[order[mode]])
    out = {}
    g_prev = 0.0
    for g in gammas:
        # continuation in small gamma steps
        steps = max(1, int(np.ceil(abs(g - g_prev) / 5.0)))
        for gi in np.linspace(g_prev, g, steps + 1)[1:]:
            for _ in range(50):
                F = A0 @ u + gi * u ** 3 - lam * u
                G = h * (u * u).sum() - 1.0
                J = (A0 + sp.diags(3.0 * gi * u * u - lam)).tocsc()
                # bordered system [[J, -u], [2h u^T, 0]] [du; dlam] = -[F; G]
                K = sp.bmat([[J, sp.csc_matrix(-u.reshape(-1, 1))], [sp.csc_matrix(2 * h * u.reshape(1, -1)), None]]).tocsc()
                d = spla.spsolve(K, -np.concatenate([F, [G]]))
                u = u + d[:-1]
                lam = lam + d[-1]
                if np.abs(d).max() < 1e-13:
                    break
        g_prev = g
        out[float(g)] = (lam, x.copy(), u.copy())
    return out


def ground_state_1d(gammas, c=1.0, vscale=1.0, half=12.0, n=4801, mode=0):
    """-> {gamma: lam} Richardson-extrapolated from grids n and 2n-1 (second-order scheme)."""
    gammas = sorted(float(g) for g in gammas)
    a = _solve_on_grid(gammas, n, half, c, vscale, mode)
    b = _solve_on_grid(gammas, 2 * n - 1, half, c, vscale, mode)
    return {g: (4.0 * b[g][0] - a[g][0]) / 3.0 for g in gammas}, {g: (b[g][1], b[g][2]) for g in gammas}
