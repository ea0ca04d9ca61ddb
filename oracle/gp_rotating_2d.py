"""
oracle/gp_rotating_2d.py -- independent fp64 ground truth for BASELINE configs[3]: the 2D ROTATING trap (complex psi, -Omega L_z),
whose ground state above the critical rotation is a vortex lattice.  TEST INFRASTRUCTURE ONLY (checker for the E / mu / density half
of the metric on cfg4; never imported by the product).  The reference has no rotating problem; its 2D residual
template is src/gross_pitaevskii_2D.py:183-195 and the rotation term is the north star's.

Stationary states of the rotating-frame energy

    E[psi] = int 1/2 |grad psi|^2 + 1/2 r^2 |psi|^2 + g/2 |psi|^4 - Omega conj(psi) L_z psi ,   L_z = -i (x d_y - y d_x),  int |psi|^2 = 1

on a periodic box with Fourier-spectral derivatives (the condensate decays like a Gaussian: images below 1e-30).  A vortex lattice
is a LOCAL minimum, one of many that differ in orientation / arrangement by ~1e-3 in E: SURVEY 8(c) therefore asks that E and mu be
compared "against the build's own grid solver started from the same vortex-seeded initial state" -- `seed_state` is that state
(Thomas-Fermi profile of the rotating trap times one phase winding per site of a triangular lattice of the Feynman density Omega/pi),
`minimise` is the solver: preconditioned nonlinear conjugate gradients on the sphere int |psi|^2 = 1 (direction = kinetic-
preconditioned residual, Polak-Ribiere, exact minimisation along the great circle psi cos(t) + d sin(t) of the quartic energy), which
converges in a few thousand iterations where imaginary time needs 1e5.  No time step, no splitting error; the grid is the only
discretisation parameter (checked at two resolutions in tests/test_ground_state_cpu.py).

Known answers (tests): Omega = 0 reproduces oracle/gp_ground_state_nd.py (mu(2D, g = 500) = 12.678319); g = 0, Omega < 1 leaves the
Gaussian with mu = 1; a single centred vortex at g = 0 has mu = 2 - Omega (the m = 1 Landau state).
"""
from __future__ import annotations

import math

import numpy as np


class Box:
    """n x n periodic grid on [-half, half)^2, spectral derivatives."""

    def __init__(self, n, half):
        self.n, self.half = int(n), float(half)
        self.h = 2.0 * self.half / self.n
        self.x = -self.half + self.h * np.arange(self.n)
        self.X, self.Y = np.meshgrid(self.x, self.x, indexing="ij")
        k = 2.0 * np.pi * np.fft.fftfreq(self.n, d=self.h)
        self.KX, self.KY = np.meshgrid(k, k, indexing="ij")
        self.k2 = self.KX ** 2 + self.KY ** 2
        self.dv = self.h * self.h
        self.V = 0.5 * (self.X ** 2 + self.Y ** 2)

    def grad(self, psi):
        f = np.fft.fft2(psi)
        return np.fft.ifft2(1j * self.KX * f), np.fft.ifft2(1j * self.KY * f)

    def lap(self, psi):
        return np.fft.ifft2(-self.k2 * np.fft.fft2(psi))

    def lz(self, psi):
        px, py = self.grad(psi)
        return -1j * (self.X * py - self.Y * px)

    def inner(self, a, b):
        return self.dv * np.vdot(a, b)


def h_apply(bx: Box, psi, g, omega, rho=None):
    rho = np.abs(psi) ** 2 if rho is None else rho
    return -0.5 * bx.lap(psi) + (bx.V + g * rho) * psi - omega * bx.lz(psi)


def energy_parts(bx: Box, psi, g, omega):
    """-> dict(kin, pot, inter, rot, E, mu, lz) of a NORMALISED psi."""
    px, py = bx.grad(psi)
    rho = np.abs(psi) ** 2
    kin = 0.5 * bx.dv * float((np.abs(px) ** 2 + np.abs(py) ** 2).sum())
    pot = bx.dv * float((bx.V * rho).sum())
    inter = 0.5 * g * bx.dv * float((rho ** 2).sum())
    lz = float(np.real(bx.inner(psi, -1j * (bx.X * py - bx.Y * px))))
    return dict(kin=kin, pot=pot, inter=inter, lz=lz, rot=-omega * lz, E=kin + pot + inter - omega * lz,
                mu=kin + pot + 2.0 * inter - omega * lz)


def tf_radius(g, omega):
    mu_tf = math.sqrt(g * (1.0 - omega ** 2) / math.pi)
    return math.sqrt(2.0 * mu_tf / (1.0 - omega ** 2)), mu_tf


def lattice_sites(g, omega, fill=1.0, rmax_frac=0.92):
    """Triangular lattice of the Feynman vortex density Omega / pi (cell area pi / Omega), one site at the centre, inside rmax_frac R_TF."""
    R, _ = tf_radius(g, omega)
    b = math.sqrt(2.0 * math.pi / (math.sqrt(3.0) * omega * fill))
    sites = []
    m = int(R / b) + 2
    for i in range(-m, m + 1):
        for j in range(-m, m + 1):
            x = b * (i + 0.5 * j)
            y = b * (math.sqrt(3.0) / 2.0) * j
            if math.hypot(x, y) <= rmax_frac * R:
                sites.append((x, y))
    return np.array(sites), b


def seed_state(bx: Box, g, omega, sites=None, core=0.35):
    """Thomas-Fermi profile of the rotating trap (smoothed edge) times prod_k (z - z_k) / sqrt(|z - z_k|^2 + core^2); normalised."""
    R, mu_tf = tf_radius(g, omega)
    r2 = bx.X ** 2 + bx.Y ** 2
    rho = np.maximum(mu_tf - 0.5 * (1.0 - omega ** 2) * r2, 0.0) / g
    amp = np.sqrt(rho + 1e-4 * np.exp(-r2 / (2.0 * (0.5 * R) ** 2)))           # a thin Gaussian skirt: no hard edge
    if sites is None:
        sites, _ = lattice_sites(g, omega)
    z = bx.X + 1j * bx.Y
    psi = amp.astype(np.complex128)
    for (sx, sy) in sites:
        w = z - (sx + 1j * sy)
        psi = psi * w / np.sqrt(np.abs(w) ** 2 + core ** 2)
    psi /= math.sqrt(bx.dv * float((np.abs(psi) ** 2).sum()))
    return psi


def seed_at(points, g, omega, sites, core=0.35):
    """The same seed evaluated at arbitrary points [N, 2] (UN-normalised amplitude / phase; the caller normalises on its own grid)."""
    R, mu_tf = tf_radius(g, omega)
    x, y = points[:, 0].astype(np.float64), points[:, 1].astype(np.float64)
    r2 = x * x + y * y
    rho = np.maximum(mu_tf - 0.5 * (1.0 - omega ** 2) * r2, 0.0) / g
    psi = np.sqrt(rho + 1e-4 * np.exp(-r2 / (2.0 * (0.5 * R) ** 2))).astype(np.complex128)
    z = x + 1j * y
    for (sx, sy) in sites:
        w = z - (sx + 1j * sy)
        psi = psi * w / np.sqrt(np.abs(w) ** 2 + core ** 2)
    return psi


def minimise(bx: Box, psi, g, omega, tol=1e-9, max_iter=20000, verbose=False, log_every=200):
    """Preconditioned nonlinear CG on the unit sphere.  -> dict(psi, E, mu, residual, iterations, history)."""
    psi = psi / math.sqrt(float(np.real(bx.inner(psi, psi))))
    d_prev = None
    pr_prev = None
    r_prev = None
    hist = []
    res = np.inf
    for it in range(max_iter):
        rho = np.abs(psi) ** 2
        Hpsi = h_apply(bx, psi, g, omega, rho)
        mu = float(np.real(bx.inner(psi, Hpsi)))
        r = Hpsi - mu * psi                                             # Riemannian gradient (x 1/2)
        res = math.sqrt(float(np.real(bx.inner(r, r))))
        if it % log_every == 0 or res < tol:
            e = energy_parts(bx, psi, g, omega)
            hist.append((it, e["E"], mu, res))
            if verbose:
                print(f"   it {it:6d}  E {e['E']:.10f}  mu {mu:.10f}  Lz {e['lz']:.5f}  |r| {res:.3e}", flush=True)
        if res < tol:
            break
        # preconditioner: (alpha - 1/2 lap)^-1 with alpha = the local potential scale of the state
        alpha = max(float(bx.dv * ((bx.V + g * rho) * rho).sum()), 1.0)
        pr = np.fft.ifft2(np.fft.fft2(r) / (alpha + 0.5 * bx.k2))
        pr = pr - bx.inner(psi, pr) * psi                               # tangent
        if d_prev is None:
            d = -pr
        else:
            beta = max(0.0, float(np.real(bx.inner(r, pr) - bx.inner(r, pr_prev))) / max(float(np.real(bx.inner(r_prev, pr_prev))), 1e-300))
            d = -pr + beta * d_prev
            d = d - bx.inner(psi, d) * psi
            if float(np.real(bx.inner(r, d))) >= 0.0:                   # not a descent direction: restart
                d = -pr
        dn = math.sqrt(float(np.real(bx.inner(d, d))))
        if dn < 1e-300:
            break
        p = d / dn
        # energy along psi(t) = cos t psi + sin t p (exactly normalised): quadratic part via <a, H0 b>, quartic via the densities
        H0psi = Hpsi - g * rho * psi
        H0p = -0.5 * bx.lap(p) + bx.V * p - omega * bx.lz(p)
        a = float(np.real(bx.inner(psi, H0psi)))
        b = float(np.real(bx.inner(p, H0p)))
        c = float(np.real(bx.inner(psi, H0p)))
        rp = np.abs(p) ** 2
        rc = 2.0 * np.real(np.conj(psi) * p)

        def e_of(t):
            ct, st = math.cos(t), math.sin(t)
            dens = ct * ct * rho + st * st * rp + ct * st * rc
            return ct * ct * a + st * st * b + 2.0 * ct * st * c + 0.5 * g * bx.dv * float((dens * dens).sum())

        # initial step from the quadratic model, then a safeguarded golden-section refinement on [0, 2 t0]
        e0 = e_of(0.0)
        slope = 2.0 * float(np.real(bx.inner(p, r)))                    # dE/dt at 0
        curv = 2.0 * (b - a) + 2.0 * g * bx.dv * float((rc * rc + 2.0 * rho * (rp - rho)).sum()) * 0.5
        t0 = -slope / curv if curv > 1e-12 else 0.1
        t0 = min(max(t0, 1e-6), 0.5)
        lo, hi = 0.0, 2.0 * t0
        gr = 0.5 * (math.sqrt(5.0) - 1.0)
        x1, x2 = hi - gr * (hi - lo), lo + gr * (hi - lo)
        f1, f2 = e_of(x1), e_of(x2)
        for _ in range(18):
            if f1 < f2:
                hi, x2, f2 = x2, x1, f1
                x1 = hi - gr * (hi - lo)
                f1 = e_of(x1)
            else:
                lo, x1, f1 = x1, x2, f2
                x2 = lo + gr * (hi - lo)
                f2 = e_of(x2)
        t = 0.5 * (lo + hi)
        if e_of(t) > e0:                                                # safeguard: shrink
            t = t0
            while e_of(t) > e0 and t > 1e-12:
                t *= 0.5
        psi = math.cos(t) * psi + math.sin(t) * p
        psi /= math.sqrt(float(np.real(bx.inner(psi, psi))))
        d_prev = d - bx.inner(psi, d) * psi                              # transported to the new tangent space by projection
        pr_prev, r_prev = pr, r
    e = energy_parts(bx, psi, g, omega)
    return dict(psi=psi, E=e["E"], mu=e["mu"], lz=e["lz"], parts=e, residual=res, iterations=it + 1, history=hist)


def count_vortices(bx: Box, psi, rho_frac=0.02):
    """Phase windings of +-2 pi around grid plaquettes where the density exceeds rho_frac of its maximum.  -> (n_plus, n_minus, sites)"""
    ph = np.angle(psi)

    def dwrap(a):
        return (a + np.pi) % (2.0 * np.pi) - np.pi

    d1 = dwrap(np.roll(ph, -1, 0) - ph)                                # (i,j) -> (i+1,j)
    d2 = dwrap(np.roll(np.roll(ph, -1, 0), -1, 1) - np.roll(ph, -1, 0))
    d3 = dwrap(np.roll(ph, -1, 1) - np.roll(np.roll(ph, -1, 0), -1, 1))
    d4 = dwrap(ph - np.roll(ph, -1, 1))
    w = np.rint((d1 + d2 + d3 + d4) / (2.0 * np.pi)).astype(int)
    rho = np.abs(psi) ** 2
    # a vortex core has (near) zero density: judge "inside the cloud" by the smoothed density around the plaquette
    k = np.fft.ifft2(np.fft.fft2(rho) * np.exp(-0.5 * bx.k2 * 0.6 ** 2)).real
    mask = k > rho_frac * k.max()
    ip, jp = np.nonzero((w > 0) & mask)
    im, jm = np.nonzero((w < 0) & mask)
    sites = np.stack([bx.x[ip] + 0.5 * bx.h, bx.x[jp] + 0.5 * bx.h], axis=1) if len(ip) else np.zeros((0, 2))
    return int(len(ip)), int(len(im)), sites


def interp_density(bx: Box, psi, pts):
    """|psi|^2 at arbitrary points [N, 2] (cubic interpolation of the periodic grid function)."""
    from scipy.interpolate import RegularGridInterpolator
    ax = np.append(bx.x, bx.half)
    rho = np.abs(psi) ** 2
    rp = np.concatenate([rho, rho[:1, :]], axis=0)
    rp = np.concatenate([rp, rp[:, :1]], axis=1)
    return RegularGridInterpolator((ax, ax), rp, method="cubic")(pts)


if __name__ == "__main__":
    import json
    import os
    import sys
    import time
    g, omega = 500.0, 0.8
    out = {}
    rows = []
    sites, b = lattice_sites(g, omega)
    for n in ([192, 256] if len(sys.argv) < 2 else [int(v) for v in sys.argv[1:]]):
        bx = Box(n, 12.0)
        t0 = time.time()
        r = minimise(bx, seed_state(bx, g, omega, sites), g, omega, tol=2e-9, verbose=True, log_every=500)
        nv = count_vortices(bx, r["psi"])
        rows.append(dict(n=n, half=12.0, E=r["E"], mu=r["mu"], lz=r["lz"], residual=r["residual"], iterations=r["iterations"],
                         vortices=nv[0], antivortices=nv[1], seconds=time.time() - t0, parts={k: float(v) for k, v in r["parts"].items()}))
        print(rows[-1], flush=True)
    out["2d_rot_g500_om0.8"] = dict(problem=dict(g=g, omega_rot=omega, omega=[1.0, 1.0]), seed=dict(sites=sites.tolist(), spacing=b, core=0.35),
                                    grids=rows, mu=rows[-1]["mu"], energy=rows[-1]["E"], lz=rows[-1]["lz"], vortices=rows[-1]["vortices"],
                                    grid_independence_mu=abs(rows[-1]["mu"] - rows[0]["mu"]), grid_independence_E=abs(rows[-1]["E"] - rows[0]["E"]))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gp_ground_truth.json")
    prev = json.load(open(path)) if os.path.exists(path) else {}
    prev.update(out)
    json.dump(prev, open(path, "w"), indent=1)
    print("wrote", path)
