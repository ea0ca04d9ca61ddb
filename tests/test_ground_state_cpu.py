"""CPU: the independent fp64 ground-state solvers (checkers of the |mu - mu_ref| half of the BASELINE metric) against known
answers and against each other.  oracle/gp_ground_truth.json holds the values the accuracy runs are judged against
(produced by `python oracle/gp_ground_state_nd.py`, two grids per case)."""
import json
import os

import numpy as np
import pytest

from oracle import gp_ground_state as gs1
from oracle import gp_ground_state_nd as nd

TRUTH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "gp_ground_truth.json")


@pytest.mark.parametrize("omega,n,half", [([1.0], [256], [12.0]), ([1.0, 1.0], [64, 64], [9.0, 9.0]),
                                           ([1.0, 1.4, 2.0], [40, 36, 32], [7.0, 6.0, 5.0])])
def test_linear_limit_is_half_sum_omega(omega, n, half):
    r = nd.ground_state(omega, 0.0, n, half)
    assert abs(r["mu"] - 0.5 * sum(omega)) < 1e-10 and abs(r["energy"] - 0.5 * sum(omega)) < 1e-10


def test_1d_spectral_newton_equals_finite_difference_newton():
    """two unrelated discretisations (Fourier-spectral / bordered MINRES vs 3-point stencil / sparse LU + Richardson)"""
    a = nd.ground_state([1.0], 100.0, [512], [16.0])["mu"]
    lam, _ = gs1.ground_state_1d([100.0], c=0.5, vscale=0.5, half=14.0, n=2401)
    assert abs(a - float(lam[100.0])) < 2e-6
    assert abs(a - 14.134287) < 2e-6                      # 14.134


def test_2d_g500_value_and_grid_independence():
    a = nd.ground_state([1.0, 1.0], 500.0, [96, 96], [9.0, 9.0])
    b = nd.ground_state([1.0, 1.0], 500.0, [128, 128], [10.0, 10.0])
    assert abs(a["mu"] - b["mu"]) < 1e-6 and b["residual"] < 1e-10
    assert abs(b["mu"] - 12.678319) < 2e-6                # 12.678 ; E = 8.5118
    assert abs(b["energy"] - 8.511845) < 2e-6
    assert abs(b["mu"] - b["mu_from_energy"]) < 1e-9      # mu = E_kin + E_pot + 2 E_int (virial-type identity of the solution)
    tf = nd.thomas_fermi_mu([1.0, 1.0], 500.0)
    assert 0 < b["mu"] - tf < 0.1 * tf                    # Thomas-Fermi from below, within its known few-percent error


def test_committed_ground_truth_is_consistent():
    t = json.load(open(TRUTH))
    for name in ("1d_g100", "2d_g500", "3d_aniso_g1000"):
        assert t[name]["grid_independence"] < 1e-5, name
        assert all(g["residual"] < 1e-9 for g in t[name]["grids"])
    assert abs(t["2d_g500"]["mu"] - 12.678319) < 2e-6
    assert abs(t["3d_aniso_g1000"]["mu"] - 13.089) < 2e-3   # SURVEY 8(c)'s scratch value, good to ~1e-3
    assert abs(t["1d_g100"]["mu"] - 14.134287) < 2e-6


# ---- the rotating-frame solver (BASELINE configs[3]: complex psi, -Omega L_z, vortex lattice) ---------------------------------------
def test_rotating_solver_known_answers():
    """oracle/gp_rotating_2d.py (preconditioned nonlinear CG on the sphere, Fourier-spectral, complex psi): Omega = 0 reproduces the
    Newton solver of oracle/gp_ground_state_nd.py; below the critical rotation the vortex-free state is unchanged (L_z psi = 0);
    at g = 0 the m = 1 Landau state has E = mu = 2 - Omega and <L_z> = 1; phase windings are counted correctly."""
    from oracle import gp_rotating_2d as R
    bx = R.Box(96, 9.0)
    psi0 = np.exp(-0.5 * (bx.X ** 2 + bx.Y ** 2) / 4.0).astype(complex)
    a = R.minimise(bx, psi0, 500.0, 0.0, tol=1e-9)
    b = nd.ground_state([1.0, 1.0], 500.0, [96, 96], [9.0, 9.0])
    assert abs(a["mu"] - b["mu"]) < 1e-7 and abs(a["E"] - b["energy"]) < 1e-7 and abs(a["lz"]) < 1e-9
    c = R.minimise(bx, psi0, 20.0, 0.3, tol=1e-9)                       # slow rotation: no vortex enters, the state does not notice
    d = nd.ground_state([1.0, 1.0], 20.0, [96, 96], [9.0, 9.0])
    assert abs(c["mu"] - d["mu"]) < 1e-7 and abs(c["lz"]) < 1e-8 and R.count_vortices(bx, c["psi"])[:2] == (0, 0)
    z = bx.X + 1j * bx.Y
    m1 = z * np.exp(-0.5 * (bx.X ** 2 + bx.Y ** 2))
    m1 /= np.sqrt(bx.dv * (np.abs(m1) ** 2).sum())
    e = R.energy_parts(bx, m1, 0.0, 0.4)
    assert abs(e["E"] - 1.6) < 1e-9 and abs(e["mu"] - 1.6) < 1e-9 and abs(e["lz"] - 1.0) < 1e-9
    off = (z - (0.37 + 0.21j)) * np.exp(-0.5 * (bx.X ** 2 + bx.Y ** 2))      # (core between grid points, as in every converged lattice)
    assert R.count_vortices(bx, off)[:2] == (1, 0) and R.count_vortices(bx, np.conj(off))[:2] == (0, 1)


def test_rotating_ground_truth_record():
    """oracle/gp_ground_truth.json: the vortex-lattice state of cfg4 from the committed seed, two grids (160^2, 224^2 on [-12,12)^2)."""
    t = json.load(open(TRUTH))["2d_rot_g500_om0.8"]
    assert t["grid_independence_mu"] < 1e-7 and t["grid_independence_E"] < 1e-7
    assert all(g["residual"] < 1e-8 and g["vortices"] == 19 and g["antivortices"] == 0 for g in t["grids"])
    assert abs(t["mu"] - 8.7315964) < 1e-6 and abs(t["energy"] - 6.0997440) < 1e-6 and abs(t["lz"] - 10.32511) < 1e-4
    # the lattice is the favourable state: the vortex-free stationary state of the same problem has a higher rotating-frame energy
    free = nd.ground_state([1.0, 1.0], 500.0, [96, 96], [9.0, 9.0])
    assert t["energy"] < free["energy"] - 1.0


def test_rotating_solver_from_the_committed_seed_small_grid():
    from oracle import gp_rotating_2d as R
    t = json.load(open(TRUTH))["2d_rot_g500_om0.8"]
    bx = R.Box(128, 12.0)
    r = R.minimise(bx, R.seed_state(bx, 500.0, 0.8, np.array(t["seed"]["sites"]), core=t["seed"]["core"]), 500.0, 0.8, tol=1e-7, max_iter=4000)
    assert abs(r["mu"] - t["mu"]) < 1e-5 and R.count_vortices(bx, r["psi"])[0] == 19
