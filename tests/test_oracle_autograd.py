"""CPU: the hand-derived jet forward / reverse pass of oracle/gpe_oracle.py against torch autograd (oracle/torch_ref.py:
the reference's own op sequence -- nn.Sequential, autograd.grad(create_graph=True) twice per coordinate, backward) in
fp64, on every flavour the reference does not contain and the HIP kernels are judged against: 2D g=500, 3D anisotropic,
complex psi + rotation, symmetry, orthogonality, Riesz.  Tolerances: loss / mu 1e-12 relative, gradient 1e-10 of max|g|.
Also checks SURVEY quirk Q10 (the lambda branch of autograd is identically zero) by running autograd with lambda attached.
"""
import numpy as np
import pytest
import torch

from oracle import gpe_oracle as go
from oracle import torch_ref as tr

CASES = {
    "1d_refine": (dict(layers=[1, 16, 16, 1], activation=1, kinetic_coeff=1.0, pot_scale=1.0, gamma=5.0, base_mode=2,
                       perturb_scale=0.07, dx=0.1), 40),
    "1d_p4_abs": (dict(layers=[1, 12, 12, 12, 1], gamma=2.0, p=4, abs_power=True, base_mode=1, dx=0.1), 33),
    "1d_sym": (dict(layers=[1, 16, 16, 1], gamma=1.0, base_mode=0, w_sym=5.0, dx=0.1), 31),
    "1d_sym_odd": (dict(layers=[1, 16, 16, 1], gamma=1.0, base_mode=1, w_sym=5.0, sym_sign=-1.0, dx=0.1), 31),
    "1d_riesz": (dict(layers=[1, 16, 16, 1], gamma=10.0, abs_power=True, base_mode=0, w_riesz=1.0, w_sym=5.0, dx=12 / 49), 50),
    "2d_g500": (dict(layers=[2, 24, 24, 24, 24, 1], gamma=500.0, dx=0.02), 50),
    "3d_aniso_g1000": (dict(layers=[3, 20, 20, 20, 1], gamma=1000.0, omega=(1.0, 1.4, 2.0), dx=0.01), 50),
    "2d_complex_rot": (dict(layers=[2, 16, 16, 16, 2], complex_psi=True, gamma=50.0, omega_rot=0.8, dx=0.02), 40),
    "3d_complex": (dict(layers=[3, 12, 12, 2], complex_psi=True, gamma=5.0, dx=0.02), 20),
    "1d_orth2": (dict(layers=[1, 16, 16, 1], gamma=3.0, base_mode=2, w_orth=7.0, dx=0.1), 41),
    "2d_orth1": (dict(layers=[2, 16, 16, 16, 1], gamma=20.0, w_orth=3.0, dx=0.03), 37),
    "2d_riesz_sum": (dict(layers=[2, 16, 16, 16, 1], gamma=100.0, kinetic_coeff=1.0, pot_scale=1.0, w_riesz=0.3, riesz_kind=go.RIESZ_SUM,
                          dx=0.02), 45),
    "2d_riesz_variational": (dict(layers=[2, 16, 16, 16, 1], gamma=500.0, w_riesz=2.0, riesz_kind=go.RIESZ_VARIATIONAL, dx=0.02), 45),
    "3d_riesz_variational_p2": (dict(layers=[3, 12, 12, 1], gamma=30.0, p=2, abs_power=True, omega=(1.0, 1.4, 2.0), w_riesz=1.5,
                                     riesz_kind=go.RIESZ_VARIATIONAL, dx=0.02), 30),
    "2d_complex_rot_variational": (dict(layers=[2, 16, 16, 16, 2], complex_psi=True, gamma=50.0, omega_rot=0.8, w_riesz=1.5,
                                        riesz_kind=go.RIESZ_VARIATIONAL, dx=0.02), 40),
    "2d_complex_variational_norot": (dict(layers=[2, 12, 12, 2], complex_psi=True, gamma=20.0, w_riesz=1.0, riesz_kind=go.RIESZ_VARIATIONAL,
                                          dx=0.02), 30),
    "2d_riesz_paper": (dict(layers=[2, 12, 12, 1], gamma=10.0, w_riesz=1.0, riesz_kind=go.RIESZ_PAPER, dx=0.02), 30),
    "1d_shifted_beta_trap": (dict(layers=[1, 16, 16, 1], activation=1, kinetic_coeff=1.0, pot_scale=0.35, omega=(3.0, 1.0, 1.0), pot_a=0.7,
                                  gamma=4.0, dx=0.1), 40),
    "1d_residual_box_gauss": (dict(layers=[1, 16, 16, 16, 1], net_kind=go.NET_RESIDUAL, activation=1, kinetic_coeff=1.0, potential=go.POT_GAUSSIAN,
                                   pot_a=0.5, gamma=3.0, p=4, base_mode=1, perturb_scale=0.05, dx=0.02), 40),
    "2d_residual_3blocks": (dict(layers=[2, 12, 12, 12, 12, 1], net_kind=go.NET_RESIDUAL, gamma=20.0, dx=0.02), 30),
    # the 2D classes' loss (src/gross_pitaevskii_2D.py:154-242): energy-functional lambda (its gradient branch does NOT vanish), the two
    # regularisers, unnormalised Riesz sum, 10 x boundary mean, no normalisation term
    "2d_class_loss": (dict(layers=[2, 16, 16, 16, 1], gamma=100.0, kinetic_coeff=1.0, pot_scale=1.0, w_norm=0.0, w_riesz=1.0,
                           riesz_kind=go.RIESZ_SUM, lambda_kind=go.LAMBDA_ENERGY, w_reg_f=1.0, w_reg_lam=1.0, dx=0.02), 45),
    "2d_energy_lambda_only": (dict(layers=[2, 12, 12, 1], gamma=30.0, lambda_kind=go.LAMBDA_ENERGY, dx=0.02), 30),
    "3d_energy_lambda_p5_variational": (dict(layers=[3, 12, 12, 1], gamma=3.0, p=5, omega=(1.0, 1.4, 2.0), lambda_kind=go.LAMBDA_ENERGY,
                                             w_reg_lam=0.7, reg_lam_eps=1e-3, w_riesz=0.5, riesz_kind=go.RIESZ_VARIATIONAL, dx=0.02), 30),
    "1d_reg_f_rayleigh": (dict(layers=[1, 16, 16, 1], gamma=2.0, base_mode=0, w_reg_f=0.5, reg_f_eps=0.05, dx=0.1), 33),
    "1d_energy_lambda_gamma0": (dict(layers=[1, 12, 12, 1], gamma=0.0, lambda_kind=go.LAMBDA_ENERGY, w_reg_lam=1.0, dx=0.1), 25),
    "1d_gauss": (dict(layers=[1, 12, 12, 1], potential=go.POT_GAUSSIAN, pot_a=0.3, gamma=1.0, dx=0.1), 25),
    "1d_periodic": (dict(layers=[1, 12, 12, 1], potential=go.POT_PERIODIC, gamma=1.0, dx=0.1), 25),
}


def _inputs(kw, N, seed=0):
    rng = np.random.default_rng(seed)
    d = kw["layers"][0]
    x = np.linspace(-3, 3, N).reshape(-1, 1) if d == 1 else rng.uniform(-2.5, 2.5, (N, d))
    flat = rng.normal(0, 1, go.param_count(kw["layers"], kw.get("net_kind", 0))) * 0.4
    x_bc = np.array([[-3.0], [3.0]]) if d == 1 else rng.uniform(-2.5, 2.5, (5, d))
    orth = None
    if kw.get("w_orth", 0.0) != 0.0:
        n_o = 2 if d == 1 else 1
        orth = rng.normal(0, 1, (n_o, N))
    return x, flat, x_bc, orth


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("detach", [True, False], ids=["lambda_detached", "lambda_attached"])
def test_oracle_matches_autograd_fp64(name, detach):
    kw, N = CASES[name]
    x, flat, x_bc, orth = _inputs(kw, N)
    pb = go.Problem(**kw)
    osc, ograd, ores = go.full_loss_and_grad(pb, flat, x, x_bc, orth=orth)
    net = tr.build_network(list(pb.layers), pb.activation, torch.float64, pb.net_kind)
    tr.set_flat(net, flat)
    X = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    total, pieces = tr.epoch_losses(pb, net, X, torch.tensor(x_bc, dtype=torch.float64), detach_lambda=detach,
                                    orth=None if orth is None else torch.tensor(orth, dtype=torch.float64))
    total.backward()
    tgrad = tr.get_flat_grad(net)
    assert abs(float(total.detach()) - osc["loss"]) <= 1e-12 * abs(osc["loss"])
    assert abs(float(pieces["lam"]) - osc["mu"]) <= 1e-12 * abs(osc["mu"])
    assert abs(float(pieces["pde"]) - osc["pde"]) <= 1e-11 * abs(osc["pde"])
    if "orth" in pieces:
        assert abs(float(pieces["orth"]) - osc["orth"]) <= 1e-12 * abs(osc["orth"]) and osc["orth"] > 0
    if "riesz" in pieces:
        assert abs(float(pieces["riesz"]) - osc["riesz"]) <= 1e-12 * abs(osc["riesz"])
    if "sym" in pieces:
        assert abs(float(pieces["sym"]) - osc["sym"]) <= 1e-12 * abs(osc["sym"])
    np.testing.assert_allclose(pieces["u"].detach().numpy(), ores["psi"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(pieces["r"].detach().numpy(), ores["residual"], rtol=0, atol=1e-10 * np.abs(ores["residual"]).max())
    # quirk Q10: with lambda attached autograd adds -2 mean(r u) dlambda = 0 up to round-off -- for the Rayleigh quotient.  The
    # energy-functional lambda of the 2D classes is not stationary: there the oracle carries the branch, and cutting it must show
    if kw.get("lambda_kind", go.LAMBDA_RAYLEIGH) == go.LAMBDA_ENERGY and detach:
        assert np.abs(tgrad - ograd).max() >= 1e-6 * np.abs(ograd).max()
        return
    assert np.abs(tgrad - ograd).max() <= 1e-10 * np.abs(ograd).max()


def test_jets_match_double_autograd_per_axis():
    """mlp_forward's d/dx_k and d2/dx_k^2 channels against autograd on a 3D net."""
    layers = [3, 10, 10, 2]
    rng = np.random.default_rng(3)
    flat = rng.normal(0, 1, go.param_count(layers)) * 0.5
    x = rng.uniform(-1, 1, (7, 3))
    jets, _ = go.mlp_forward(go.unflatten(flat, layers), x, 1)
    net = tr.build_network(layers, 1, torch.float64)
    tr.set_flat(net, flat)
    X = torch.tensor(x, requires_grad=True)
    out = net(X)
    # ShiftedTanh adds eps on top of +1: invisible beyond 2e-16 per layer
    for o in range(2):
        g1 = torch.autograd.grad(out[:, o].sum(), X, create_graph=True)[0]
        np.testing.assert_allclose(jets[0][:, o], out[:, o].detach().numpy(), atol=1e-13)
        for k in range(3):
            np.testing.assert_allclose(jets[1 + k][:, o], g1[:, k].detach().numpy(), atol=1e-13)
            g2 = torch.autograd.grad(g1[:, k].sum(), X, retain_graph=True)[0]
            np.testing.assert_allclose(jets[4 + k][:, o], g2[:, k].detach().numpy(), atol=1e-12)


def test_orthogonality_definition_and_dp_additivity():
    """L_orth = sum_j (dx * sum_m psi_j u)^2 is a function of GLOBAL sums: two shards, phase-1 sums added, phase-2 gradients
    added, must equal the full-batch value (the S_ORTH sums ride in the first exchange of the engine)."""
    kw = dict(layers=[1, 16, 16, 1], gamma=3.0, base_mode=0, w_orth=11.0, dx=0.1)
    N = 60
    x, flat, x_bc, _ = _inputs(kw, N)
    rng = np.random.default_rng(5)
    orth = rng.normal(0, 1, (2, N))
    pb = go.Problem(**kw, n_global=N)
    sc, g, res = go.full_loss_and_grad(pb, flat, x, x_bc, orth=orth)
    u = res["psi"][:, 0]
    want = sum((pb.dx * float((orth[j] * u).sum())) ** 2 for j in range(2))
    assert abs(sc["orth"] - want) <= 1e-13 * want
    lo = 23
    parts = [(x[:lo], orth[:, :lo]), (x[lo:], orth[:, lo:])]
    s1 = [go.loss_and_grad(pb, flat, xs, x_bc, orth=os_, phase=1) for xs, os_ in parts]
    tot = {k: s1[0][k] + s1[1][k] for k in s1[0]}
    r2 = [go.loss_and_grad(pb, flat, xs, x_bc, orth=os_, shard_sums=tot) for xs, os_ in parts]
    gsum = r2[0]["grad_local"] + r2[1]["grad_local"] + r2[0]["grad_bc"]
    assert np.abs(gsum - g).max() <= 1e-12 * np.abs(g).max()
    assert abs(r2[0]["L_orth"] - sc["orth"]) <= 1e-13 * sc["orth"]


def test_sharded_oracle_with_the_2d_class_loss_and_a_precomputed_potential():
    """The sharded evaluation with the terms whose reverse pass needs global sums beyond (num, den): energy-functional lambda, regularisers,
    Riesz sum; potential handed over as an array."""
    kw = dict(layers=[2, 16, 16, 1], gamma=100.0, kinetic_coeff=1.0, potential=go.POT_PRECOMPUTED, abs_power=True, w_norm=0.0, w_riesz=1.0,
              riesz_kind=go.RIESZ_SUM, lambda_kind=go.LAMBDA_ENERGY, w_reg_f=1.0, w_reg_lam=1.0, dx=1.0)
    rng = np.random.default_rng(5)
    N = 1000
    x = rng.uniform(0, np.pi, (N, 2))
    V = np.exp(-((x[:, 0] - np.pi / 2) ** 2 + (x[:, 1] - np.pi / 2) ** 2) / 0.5)
    xb = rng.uniform(0, np.pi, (7, 2))
    flat = rng.normal(0, 0.4, go.param_count(kw["layers"]))
    pb = go.Problem(**kw)
    sc, g, _ = go.full_loss_and_grad(pb, flat, x, xb, V_pre=V)
    sc2, g2 = go.sharded_loss_and_grad(pb, flat, x, xb, chunk=192, threads=2, V_pre=V)
    for k in ("loss", "mu", "pde", "riesz", "reg", "bc"):
        assert abs(sc[k] - sc2[k]) <= 1e-12 * abs(sc[k]), k
    assert np.abs(g - g2).max() <= 1e-11 * np.abs(g).max()


def test_sharded_two_phase_oracle_equals_the_single_call():
    """oracle.sharded_loss_and_grad (what bench.py's in-run parity_check and the 540 000-point GPU test run on the batch as timed):
    chunked two-phase evaluation == one call on the whole batch, to the order of the fp64 sums."""
    rng = np.random.default_rng(5)
    layers = [2, 32, 32, 32, 1]
    pb = go.Problem(layers=layers, gamma=300.0, p=3, kinetic_coeff=0.5, pot_scale=0.5, dx=2e-3, w_bc=10.0, w_norm=20.0)
    x = rng.uniform(-4, 4, (3001, 2))
    xb = rng.uniform(-4, 4, (17, 2))
    flat = rng.normal(0, 0.3, go.param_count(layers))
    sc, g, _ = go.full_loss_and_grad(pb, flat, x, xb)
    sc2, g2 = go.sharded_loss_and_grad(pb, flat, x, xb, chunk=700, threads=3)
    for k in ("loss", "pde", "bc", "norm", "mu"):
        assert abs(sc[k] - sc2[k]) <= 1e-12 * max(1.0, abs(sc[k])), k
    assert np.abs(g - g2).max() <= 1e-12 * np.abs(g).max()
