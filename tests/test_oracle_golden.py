"""Pin the CPU oracle (oracle/gpe_oracle.py, oracle/torch_ref.py) against golden vectors produced by the
imported reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import gpe_oracle as go
from tests import helpers as H


@pytest.mark.parametrize("name", H.refine_names())
def test_refine_oplevel(name):
    fx = H.load_fx(name)
    pb = H.problem_from_refine(fx)
    x = fx["x"].astype(np.float64)
    flat = fx["flat0"].astype(np.float64)
    params = go.unflatten(flat, pb.layers)
    out, _ = go.mlp_forward(params, x, pb.activation)
    assert H.rel_err(out[0], fx["nn_out"]) < 1e-5                       # forward (a3)
    h = go.head_pde(pb, x, out)
    U = h["U"]
    assert H.rel_err(U[0], fx["u"]) < 2e-6                              # a5
    assert H.rel_err(U[1], fx["u_x"]) < 5e-6                            # K2: autograd d/dx
    assert H.rel_err(U[2], fx["u_xx"]) < 2e-5                           # K3: autograd d2/dx2
    assert H.rel_err(h["V"][:, None], fx["V"]) < 1e-6                   # a6
    sc, grad, res = go.full_loss_and_grad(pb, flat, x, H.bc_points(fx))
    assert abs(sc["mu"] - float(fx["lam"])) < 2e-5 * max(1.0, abs(float(fx["lam"])))   # K6
    assert H.rel_err(res["residual"], fx["residual"]) < 1e-4            # K7 (residual is a difference of O(1) terms)
    assert abs(sc["pde"] - float(fx["pde_loss"])) < 2e-4 * max(1e-3, float(fx["pde_loss"]))
    assert abs(sc["bc"] - float(fx["bc_loss"])) < 1e-5 * max(1e-6, float(fx["bc_loss"])) + 1e-12
    assert abs(sc["norm"] - float(fx["norm_loss"])) < 1e-4 * max(1e-3, float(fx["norm_loss"]))
    assert abs(sc["loss"] - float(fx["total"])) < 1e-4 * max(1e-3, float(fx["total"]))
    # K12: d total / d theta (reference: autograd through the double-backward graph, fp32)
    assert H.rel_err(grad, fx["grad0"]) < 2e-4


@pytest.mark.parametrize("name", ["fx_vanilla_m0_g10_64x3.npz", "fx_vanilla_m0_gm4_32x3.npz"])
def test_vanilla_pinn_branch_oplevel(name):
    """use_perturbation=False (refine/harmonic_pinn_simulation.py:152-155, :205-208; tests/golden/make_golden_refine_variants.py):
    u = the scaled network output, no Hermite base; also with an attractive interaction (gamma < 0)."""
    fx = H.load_fx(name)
    pb = H.problem_from_vanilla(fx)
    x = fx["x"].astype(np.float64)
    flat = fx["flat0"].astype(np.float64)
    out, _ = go.mlp_forward(go.unflatten(flat, pb.layers), x, pb.activation)
    assert H.rel_err(out[0], fx["nn_out"]) < 1e-5
    h = go.head_pde(pb, x, out)
    assert H.rel_err(h["U"][0], fx["u"]) < 2e-6 and H.rel_err(h["U"][1], fx["u_x"]) < 5e-6 and H.rel_err(h["U"][2], fx["u_xx"]) < 2e-5
    sc, grad, res = go.full_loss_and_grad(pb, flat, x, H.bc_points(fx))
    assert abs(sc["mu"] - float(fx["lam"])) < 2e-5 * abs(float(fx["lam"]))
    assert H.rel_err(res["residual"], fx["residual"]) < 1e-4
    assert abs(sc["pde"] - float(fx["pde_loss"])) < 2e-4 * float(fx["pde_loss"])
    assert abs(sc["bc"] - float(fx["bc_loss"])) < 1e-5 * float(fx["bc_loss"]) + 1e-12
    assert abs(sc["norm"] - float(fx["norm_loss"])) < 1e-4 * float(fx["norm_loss"])
    assert abs(sc["loss"] - float(fx["total"])) < 1e-4 * float(fx["total"])
    assert H.rel_err(grad, fx["grad0"]) < 2e-4


@pytest.mark.parametrize("name", H.nb_names())
def test_notebook_oplevel(name):
    fx = H.load_fx(name)
    pb = H.problem_from_nb(fx)
    x = fx["x"].astype(np.float64)
    flat = fx["flat0"].astype(np.float64)
    sc, grad, res = go.full_loss_and_grad(pb, flat, x, H.bc_points(fx))
    assert H.rel_err(res["psi"], fx["u"]) < 2e-6
    assert abs(sc["mu"] - float(fx["lam"])) < 5e-5 * max(1.0, abs(float(fx["lam"])))
    assert H.rel_err(res["residual"], fx["residual"]) < 2e-4
    assert abs(sc["pde"] - float(fx["pde_loss"])) < 5e-4 * float(fx["pde_loss"])
    assert abs(sc["sym"] - float(fx["sym_loss"])) < 1e-4 * max(1e-6, float(fx["sym_loss"]))
    assert abs(sc["loss"] - float(fx["total"])) < 2e-4 * float(fx["total"])
    assert H.rel_err(grad, fx["grad0"]) < 5e-4


@pytest.mark.parametrize("name", ["fx_refine_m0_g0_64x3.npz", "fx_refine_m0_g50_64x3.npz", "fx_refine_m2_g10_32x4.npz"])
def test_refine_trace_fp32(name):
    """(loss, mu, lr) trace of the reference loop body (Adam + clip + cosine(loss) scheduler, quirk Q4)."""
    fx = H.load_fx(name)
    pb = H.problem_from_refine(fx)
    st = go.OptState(lr0=1e-3, sched=go.SCHED_COSINE_LOSS)
    n = 25
    flat, trace = go.train_steps(pb, st, fx["flat0"], fx["x"], n, x_bc=H.bc_points(fx))
    loss = np.array([t["loss"] for t in trace]); mu = np.array([t["mu"] for t in trace])
    lr = np.array([t["lr"] for t in trace]); gn = np.array([t["grad_norm"] for t in trace])
    np.testing.assert_allclose(lr, fx["trace_lr"][:n], rtol=2e-3, atol=1e-9)
    np.testing.assert_allclose(loss[:10], fx["trace_loss"][:10], rtol=2e-3)
    np.testing.assert_allclose(mu[:10], fx["trace_mu"][:10], rtol=5e-4)
    np.testing.assert_allclose(gn[:5], fx["trace_gnorm"][:5], rtol=2e-3)
    np.testing.assert_allclose(loss, fx["trace_loss"][:n], rtol=5e-2)
    np.testing.assert_allclose(mu, fx["trace_mu"][:n], rtol=5e-3)
    # params after 1..3 optimiser steps (fp32 Adam; sign(g)-like first step is sensitive where g~0)
    f1, _ = go.train_steps(pb, go.OptState(lr0=1e-3, sched=go.SCHED_COSINE_LOSS), fx["flat0"], fx["x"], 1,
                           x_bc=H.bc_points(fx))
    d = np.abs(f1 - fx["flat_after_1"])
    assert np.quantile(d, 0.99) < 2e-5 and d.max() <= 2.1e-3


@pytest.mark.parametrize("name", ["fx_nb_m0_g1_p3_32x4.npz", "fx_nb_m0_g1_p2_64x3.npz"])
def test_notebook_trace_fp32(name):
    fx = H.load_fx(name)
    pb = H.problem_from_nb(fx)
    st = go.OptState(lr0=1e-3, sched=go.SCHED_PLATEAU)
    n = 20
    flat, trace = go.train_steps(pb, st, fx["flat0"], fx["x"], n, x_bc=H.bc_points(fx))
    loss = np.array([t["loss"] for t in trace]); mu = np.array([t["mu"] for t in trace])
    np.testing.assert_allclose(loss[:8], fx["trace_loss"][:8], rtol=3e-3)
    np.testing.assert_allclose(mu[:8], fx["trace_mu"][:8], rtol=2e-3)
    np.testing.assert_allclose(loss, fx["trace_loss"][:n], rtol=8e-2)


def test_cosine_loss_schedule_closed_form():
    fx = H.load_fx("fx_cosine_loss_lr.npz")
    # the reference passes an fp32 loss tensor as the epoch, so T_cur is fp32 arithmetic there
    got = np.array([go.cosine_lr_from_loss(float(np.float32(L)), 1e-3, 200.0, 2.0, 1e-6) for L in fx["loss"]])
    np.testing.assert_allclose(got, fx["lr"], rtol=5e-6, atol=1e-12)


@pytest.mark.parametrize("name", H.refine_names())
def test_eval_density_refine(name):
    fx = H.load_fx(name)
    pb = H.problem_from_refine(fx)
    xt = fx["eval_x"].astype(np.float64)
    u, dens = go.eval_density(pb, fx["flat_final"].astype(np.float64), xt, float(xt[1, 0] - xt[0, 0]),
                              abs_mode0=(int(fx["mode"]) == 0))
    assert H.rel_err(u[:, 0], fx["eval_u"]) < 5e-5   # reference evaluates in fp32


def test_eval_density_notebook():
    fx = H.load_fx("fx_nb_m0_g1_p3_32x4.npz")
    pb = H.problem_from_nb(fx)
    xt = fx["eval_x"].astype(np.float64)
    u, dens = go.eval_density(pb, fx["flat_final"].astype(np.float64), xt, float(xt[1, 0] - xt[0, 0]))
    assert H.rel_err(dens, fx["eval_density"]) < 1e-4


def test_q1_documented():
    """The 2D reference residual broadcasts to [N,N] (quirk Q1) -- the build follows the intended [N,1] formula."""
    fx = H.load_fx("fx_q1_2d_shape.npz")
    assert tuple(fx["residual_shape"]) == (7, 7)


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
def test_stored_reference_checkpoint_mu_table(mode):
    """Known answer from the reference's own stored artefact (harmonic_mode_zero_plot_data.pkl): its trained weights must
    reproduce the lambda it recorded (1, 3, 5, 7, 9, 11 to ~1e-5) through the oracle's forward path."""
    fx = H.load_fx("fx_ckpt_harmonic_modes.npz")
    N = int(fx["N"])
    x = np.linspace(float(fx["lb"]), float(fx["ub"]), N).reshape(-1, 1)
    pb = go.Problem(layers=[int(v) for v in fx["layers"]], activation=1, kinetic_coeff=1.0, pot_scale=1.0, gamma=0.0, p=3,
                    base_mode=mode, perturb_scale=float(fx["perturb_const"]) / float(fx[f"const_mode{mode}"]),
                    dx=float(x[1, 0] - x[0, 0]), w_bc=0.0)
    res = go.loss_and_grad(pb, fx[f"flat_mode{mode}"].astype(np.float64), x, want_grad=False)
    assert abs(res["lam"] - float(fx[f"mu_mode{mode}"])) < 3e-5
    assert abs(res["lam"] - (2 * mode + 1)) < 5e-5


@pytest.mark.parametrize("name", ["fx_box_m0_g0.npz", "fx_box_m1_g20.npz"])
def test_box_oplevel(name):
    """Row f3: refine/box_pinn_simulation.py (sine base, hard boundary factor sin(pi x)) vs the oracle."""
    fx = H.load_fx(name)
    pb = H.problem_from_box(fx)
    x = fx["x"].astype(np.float64)
    flat = fx["flat0"].astype(np.float64)
    sc, grad, res = go.full_loss_and_grad(pb, flat, x, np.array([[0.0], [1.0]]))
    h = go.head_pde(pb, x, go.mlp_forward(go.unflatten(flat, pb.layers), x, pb.activation)[0])
    assert H.rel_err(h["U"][0], fx["u"]) < 2e-6
    assert H.rel_err(h["U"][1], fx["u_x"]) < 1e-5
    assert H.rel_err(h["U"][2], fx["u_xx"]) < 1e-4
    assert abs(sc["mu"] - float(fx["lam"])) < 5e-5 * abs(float(fx["lam"]))
    assert abs(sc["pde"] - float(fx["pde_loss"])) < 1e-3 * max(float(fx["pde_loss"]), 1e-4)
    assert abs(sc["loss"] - float(fx["total"])) < 1e-3 * max(float(fx["total"]), 1e-4)
    assert H.rel_err(grad, fx["grad0"]) < 1e-3


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
def test_stored_box_checkpoint_mu_table(mode):
    """box_test/box_mode_zero_plot_data.pkl: stored weights reproduce the recorded lambda = ((n+1) pi)^2."""
    fx = H.load_fx("fx_ckpt_box_modes.npz")
    N = int(fx["N"])
    x = np.linspace(0.0, 1.0, N).reshape(-1, 1)
    pb = H.problem_from_box(fx, mode=mode, const=float(fx[f"const_mode{mode}"]))
    res = go.loss_and_grad(pb, fx[f"flat_mode{mode}"].astype(np.float64), x, want_grad=False)
    ref = float(fx[f"mu_mode{mode}"])
    assert abs(res["lam"] - ref) < 2e-5 * ref
    assert abs(res["lam"] - ((mode + 1) * np.pi) ** 2) < 1e-4 * ref


@pytest.mark.parametrize("name", ["fx_gravity_m0_g0.npz", "fx_gravity_m1_g5.npz"])
def test_gravity_well_oplevel(name):
    """Row f3: refine/gravity_well_pinn_simulation.py (Airy base computed on the host with scipy there, arrays here)."""
    fx = H.load_fx(name)
    pb = H.problem_from_gravity(fx)
    x = fx["x"].astype(np.float64)
    base = (fx["base"][:, 0], fx["base_x"][:, 0], fx["base_xx"][:, 0])
    xb = np.array([[float(fx["lb"])], [float(fx["ub"])]])
    sc, grad, res = go.full_loss_and_grad(pb, fx["flat0"].astype(np.float64), x, xb, bc_target=-fx["base_boundary"],
                                          V_pre=x[:, 0], base_pre=base)
    assert H.rel_err(res["psi"], fx["u"]) < 2e-6
    assert abs(sc["mu"] - float(fx["lam"])) < 5e-5 * abs(float(fx["lam"]))
    assert abs(sc["bc"] - float(fx["bc_loss"])) < 1e-4 * max(float(fx["bc_loss"]), 1e-8)
    assert abs(sc["loss"] - float(fx["total"])) < 5e-4 * float(fx["total"])
    assert H.rel_err(grad, fx["grad0"]) < 1e-3


@pytest.mark.parametrize("name", ["fx_paper_g10_p3.npz", "fx_paper_g2_p2.npz"])
def test_paper_notebook_riesz_oplevel(name):
    """Row f4: the Riesz-energy term of the Paper notebook (cell 6:L133-183) and its mode-0 loss (cell 8:L103-142)."""
    fx = H.load_fx(name)
    pb = H.problem_from_paper(fx)
    sc, grad, res = go.full_loss_and_grad(pb, fx["flat0"].astype(np.float64), fx["x"].astype(np.float64), H.bc_points(fx))
    assert abs(sc["riesz"] - float(fx["riesz"])) < 2e-5 * abs(float(fx["riesz"]))
    assert abs(sc["pde"] - float(fx["pde_loss"])) < 5e-4 * float(fx["pde_loss"])
    assert abs(sc["loss"] - float(fx["total"])) < 2e-4 * float(fx["total"])
    assert H.rel_err(grad, fx["grad0"]) < 5e-4


# ---- d > 1 pinned by the reference's own 2D class, one point per call (quirk Q1 inert at N = 1) -----------------------------
FX_2D = ["fx_2d_ref_points_64x4_g500.npz", "fx_2d_ref_points_100x3_g100.npz", "fx_2d_ref_points_128x5_g500.npz"]


@pytest.mark.parametrize("name", FX_2D)
def test_2d_jets_and_Hu_against_reference_points(name):
    """tests/golden/make_golden_2d.py: src/gross_pitaevskii_2D_minimal.py:170-182 evaluated one point at a time.
    Oracle (fp64) jets vs the reference's fp32 autograd u_x, u_y, u_xx, u_yy; H u = -lap u + V u + g u^3 vs the reference's
    residual + lambda_pde u; and the reference's energy-functional lambda_pde = (|grad u|^2 + V u^2 + g u^4)/u^2."""
    fx = H.load_fx(name)
    layers = [int(v) for v in fx["layers"]]
    g = float(fx["g"])
    x = fx["x"].astype(np.float64)
    flat = fx["flat0"].astype(np.float64)
    jets, _ = go.mlp_forward(go.unflatten(flat, layers), x, 0)
    assert H.rel_err(jets[0][:, 0], fx["u"]) < 2e-6
    assert H.rel_err(jets[1][:, 0], fx["u_x"]) < 5e-6 and H.rel_err(jets[2][:, 0], fx["u_y"]) < 5e-6
    assert H.rel_err(jets[3][:, 0], fx["u_xx"]) < 2e-5 and H.rel_err(jets[4][:, 0], fx["u_yy"]) < 2e-5
    # the oracle's head with the reference's V fed as a precomputed potential, kinetic coefficient 1, gamma u^3
    pb = go.Problem(layers=layers, kinetic_coeff=1.0, potential=go.POT_PRECOMPUTED, gamma=g, p=3)
    h = go.head_pde(pb, x, jets, V_pre=fx["V"])
    Hu_ref = fx["residual"] + fx["lam"] * fx["u"]
    assert H.rel_err(h["Hu"][:, 0], Hu_ref) < 2e-5
    # Riesz energy of the same class on all points at once (lines 115-146): 1/2 (sum |grad u|^2 + sum V u^2 + g/2 sum u^4)
    pbr = go.Problem(layers=layers, kinetic_coeff=1.0, potential=go.POT_PRECOMPUTED, gamma=g, p=3, w_riesz=1.0, riesz_kind=go.RIESZ_SUM,
                     w_bc=0.0)
    sc, _, _ = go.full_loss_and_grad(pbr, flat, x, V_pre=fx["V"])
    assert abs(sc["riesz"] - float(fx["riesz_all"])) < 2e-6 * abs(float(fx["riesz_all"]))
    u = jets[0][:, 0]
    lam_o = (jets[1][:, 0] ** 2 + jets[2][:, 0] ** 2 + fx["V"] * u ** 2 + g * u ** 4) / u ** 2
    big = np.abs(fx["u"]) > 1e-2            # the quotient is ill-conditioned where u ~ 0
    assert np.abs(lam_o[big] / fx["lam"][big] - 1).max() < 2e-4


@pytest.mark.parametrize("name", H.CLASS2D)
def test_2d_class_loss_and_gradient_against_reference_points(name):
    """tests/golden/make_golden_2d_class.py: model.total_loss(x_k, x_bc, u_bc) of the reference's 2D class with ONE collocation point per call
    (src/gross_pitaevskii_2D_minimal.py:201-222 = src/gross_pitaevskii_2D.py:215-242) and its loss.backward() gradient, against the fp64 oracle
    with the energy-functional lambda (its gradient branch kept), both regularisers, the unnormalised Riesz sum and 10 x the boundary mean."""
    fx = H.load_fx(name)
    pb = H.problem_from_class2d(fx, n_global=1)
    flat = fx["flat0"].astype(np.float64)
    xb = fx["x_bc"].astype(np.float64)
    worst = 0.0
    for k in range(fx["x"].shape[0]):
        x = fx["x"][k:k + 1].astype(np.float64)
        sc, grad, res = go.full_loss_and_grad(pb, flat, x, xb, V_pre=H.gaussian_2d(x))
        assert abs(res["psi"][0, 0] - fx["u"][k]) < 2e-6 * max(1.0, abs(fx["u"][k]))
        assert abs(sc["mu"] / fx["lam"][k] - 1) < 2e-5, (k, sc["mu"], fx["lam"][k])
        assert abs(sc["riesz"] / fx["riesz"][k] - 1) < 2e-5
        assert abs(10.0 * sc["bc"] / fx["bc_loss"][k] - 1) < 2e-5
        assert abs((sc["pde"] + sc["reg"]) / fx["pde_loss"][k] - 1) < 5e-5, (k, sc["pde"], sc["reg"], fx["pde_loss"][k])
        assert abs(sc["loss"] / fx["total"][k] - 1) < 2e-5
        worst = max(worst, H.rel_err(grad, fx["grad"][k]))
    assert worst < 5e-5, worst          # (measured 7e-6: the fixture is fp32 autograd)


def test_2d_class_training_set_and_seeded_init_bit_exact():
    """prepare_training_data of src/gross_pitaevskii_2D_minimal.py:225-261 under np.random.seed, and torch.manual_seed + GrossPitaevskiiPINN(layers)
    + model.apply(initialize_weights) (:264-275), reproduced bit for bit by the drop-in module."""
    import torch
    import gpe_pinn
    fx = H.load_fx("fx_2d_class_data.npz")
    np.random.seed(int(fx["seed"]))
    X_f, X_u, u = gpe_pinn.pinn2d_minimal.prepare_training_data(int(fx["N_u"]), int(fx["N_f"]))
    np.testing.assert_array_equal(X_f, fx["square_X_f"])
    np.testing.assert_array_equal(X_u, fx["square_X_u"])
    np.testing.assert_array_equal(u, fx["square_u"])
    np.random.seed(5)                                      # the polar flavour (src/gross_pitaevskii_2D.py:277-295; that script needs pyDOE and
    Xp, Xu, _ = gpe_pinn.pinn2d.prepare_training_data(9, 200)     # cannot be imported here): same stream as pair-at-a-time draws, inside the disk
    np.random.seed(5)
    pairs = [(np.random.uniform(0, 2 * np.pi), np.random.uniform(0, np.pi / 2)) for _ in range(200)]
    np.testing.assert_array_equal(Xp, np.array([[np.pi / 2 + r * np.cos(a), np.pi / 2 + r * np.sin(a)] for a, r in pairs]))
    assert Xp.shape == (200, 2) and np.all((Xp[:, 0] - np.pi / 2) ** 2 + (Xp[:, 1] - np.pi / 2) ** 2 <= (np.pi / 2) ** 2 + 1e-12)
    torch.manual_seed(int(fx["init_seed"]))
    m = gpe_pinn.pinn2d.GrossPitaevskiiPINN([int(v) for v in fx["init_layers"]])
    m.apply(gpe_pinn.pinn2d.initialize_weights)
    np.testing.assert_array_equal(m._flat, fx["init_flat"])
    for fxn, seed in (("fx_2d_class_32x2_g100.npz", 0), ("fx_2d_class_64x4_g500.npz", 1)):
        f2 = H.load_fx(fxn)
        torch.manual_seed(seed)
        m = gpe_pinn.pinn2d_minimal.GrossPitaevskiiPINN([int(v) for v in f2["layers"]], g=float(f2["g"]))
        m.apply(gpe_pinn.pinn2d_minimal.initialize_weights)
        np.testing.assert_array_equal(m._flat, f2["flat0"])


# ---- a13: seeded initialisation equals the fixtures' flat0 bit for bit -------------------------------------------------------
def test_advanced_initialization_bit_exact():
    """surface._advanced_init after torch.manual_seed(seed) + default nn.Linear init == the reference's
    model.apply(advanced_initialization) (refine/harmonic_pinn_simulation.py:636-647 ; nb c18), both flavours."""
    import torch
    from gpe_pinn import surface
    seeds = {"fx_refine_m0_g0_64x3.npz": 0, "fx_refine_m0_g50_64x3.npz": 1, "fx_refine_m2_g10_32x4.npz": 2,
             "fx_refine_m5_g4_p4_64x4.npz": 3, "fx_nb_m0_g1_p3_32x4.npz": 0, "fx_nb_m0_g100_p3_64x4.npz": 1,
             "fx_nb_m0_g1_p2_64x3.npz": 2}
    for name, seed in seeds.items():
        fx = H.load_fx(name)
        layers = [int(v) for v in fx["layers"]]
        mode = int(fx["mode"])
        torch.manual_seed(seed)
        flavour = surface.refine if "refine" in name else surface.notebook
        flat = surface.seeded_reference_init(layers, mode, flavour.KIND)
        np.testing.assert_array_equal(flat, fx["flat0"], err_msg=name)


def test_box_to_gaussian_seeded_construction_bit_exact():
    """The residual-block flavour: torch.manual_seed(seed); GrossPitaevskiiPINN(layers, ...); model.apply(advanced_initialization)
    gives the reference's weights bit for bit -- its constructor builds the network twice (refine/box_to_gaussian_pinn_simulation.py:90,
    :98), i.e. two default-init draws precede the Xavier draw."""
    import torch
    import gpe_pinn
    for name, seed in (("fx_box2gauss_m0_g0.npz", 0), ("fx_box2gauss_m1_g5_p4.npz", 3)):
        fx = H.load_fx(name)
        layers = [int(v) for v in fx["layers"]]
        mode = int(fx["mode"])
        torch.manual_seed(seed)
        model = gpe_pinn.box_to_gaussian.GrossPitaevskiiPINN(layers, mode=mode, gamma=float(fx["gamma"]), L=float(fx["ub"]), use_residual=True)
        model.apply(lambda m_: gpe_pinn.box_to_gaussian.advanced_initialization(m_, mode))
        flat = np.concatenate([np.asarray(v, np.float32).ravel() for v in model.state_dict().values()])
        np.testing.assert_array_equal(flat, fx["flat0"], err_msg=name)
        assert list(model.state_dict().keys()) == [str(k) for k in fx["state_dict_keys"]]


# ---- row f3: residual-block network of refine/box_to_gaussian_pinn_simulation.py (tests/golden/make_golden_box2gauss.py) ------------
@pytest.mark.parametrize("name", ["fx_box2gauss_m0_g0.npz", "fx_box2gauss_m1_g5_p4.npz"])
def test_box_to_gaussian_residual_network_oplevel(name):
    fx = H.load_fx(name)
    pb = H.problem_from_box2gauss(fx)
    x = fx["x"].astype(np.float64)
    flat = fx["flat0"].astype(np.float64)
    assert flat.size == go.param_count(pb.layers, pb.net_kind)
    _, skip, plain = go.expand_layers(pb.layers, pb.net_kind)
    out, _ = go.mlp_forward(go.unflatten(flat, pb.layers, pb.net_kind), x, pb.activation, skip=skip, plain_tanh=plain)
    assert H.rel_err(out[0], fx["forward_out"]) < 1e-5
    h = go.head_pde(pb, x, out)
    assert H.rel_err(h["U"][0], fx["u"]) < 2e-6 and H.rel_err(h["U"][1], fx["u_x"]) < 1e-5 and H.rel_err(h["U"][2], fx["u_xx"]) < 1e-4
    assert H.rel_err(h["V"][:, None], fx["V"]) < 1e-6
    sc, grad, _ = go.full_loss_and_grad(pb, flat, x, np.array([[float(fx["lb"])], [float(fx["ub"])]]))
    assert abs(sc["mu"] - float(fx["lam"])) < 5e-5 * abs(float(fx["lam"]))
    assert abs(sc["pde"] - float(fx["pde_loss"])) < 1e-3 * float(fx["pde_loss"])
    assert abs(sc["loss"] - float(fx["total"])) < 1e-3 * float(fx["total"])
    assert H.rel_err(grad, fx["grad0"]) < 1e-3


# ---- row f3, beta-sweep flavours: refine/vary_potential_parameter_{harmonic,gravity_well,box_and_gaussian}.py -----------------------
@pytest.mark.parametrize("name", H.vbeta_names())
def test_vary_beta_oplevel(name):
    """GrossPitaevskiiPINN(layers, ..., beta[, L]).pde_loss(inputs, predictions, gamma, beta, p, ...) and the epoch-0 body of
    train_gpe_model(gamma, beta_values, ...) of the three scripts (fixtures from the imported classes) vs the oracle."""
    fx = H.load_fx(name)
    pb, arr = H.problem_from_vbeta(fx)
    x = fx["x"].astype(np.float64)
    xb = np.array([[float(fx["lb"])], [float(fx["ub"])]])
    sc, grad, res = go.full_loss_and_grad(pb, fx["flat0"].astype(np.float64), x, xb, bc_target=arr.get("bc_target"),
                                          V_pre=arr.get("V_pre"), base_pre=arr.get("base_pre"))
    assert H.rel_err(res["psi"], fx["u"]) < 2e-6
    if str(fx["flavour"]) == "harmonic":                      # the potential the class returns for this beta
        assert H.rel_err(go.potential(pb, x), fx["V"][:, 0]) < 1e-6
    assert abs(sc["mu"] - float(fx["lam"])) < 5e-5 * abs(float(fx["lam"]))
    assert abs(sc["pde"] - float(fx["pde_loss"])) < 1e-3 * max(float(fx["pde_loss"]), 1e-4)
    assert abs(sc["bc"] - float(fx["bc_loss"])) < 1e-4 * max(float(fx["bc_loss"]), 1e-8)
    assert abs(sc["loss"] - float(fx["total"])) < 1e-3 * max(float(fx["total"]), 1e-4)
    assert H.rel_err(grad, fx["grad0"]) < 1e-3


def test_vary_beta_seeded_construction_bit_exact():
    """torch.manual_seed(seed); GrossPitaevskiiPINN(layers, mode=, beta=[, L=]); model.apply(advanced_initialization) of the three
    beta-sweep scripts (their initialiser: ten times smaller gain and bias 1e-4 from mode 3 on, :788-798) -- bit for bit."""
    import torch
    import gpe_pinn
    ns = {"harmonic": gpe_pinn.vary_beta_harmonic, "gravity": gpe_pinn.vary_beta_gravity_well, "boxgauss": gpe_pinn.vary_beta_box_and_gaussian}
    for name, seed in H.VBETA_SEEDS.items():
        fx = H.load_fx(name)
        flavour, mode = str(fx["flavour"]), int(fx["mode"])
        layers = [int(v) for v in fx["layers"]]
        torch.manual_seed(seed)
        kw = dict(L=float(fx["ub"])) if flavour != "gravity" else {}
        model = ns[flavour].GrossPitaevskiiPINN(layers, mode=mode, beta=float(fx["beta"]), **kw)
        model.apply(lambda m_: ns[flavour].advanced_initialization(m_, mode))
        np.testing.assert_array_equal(model._flat, fx["flat0"], err_msg=name)
    with pytest.raises(ValueError, match="Unknown potential type"):          # the driver's default potential_type='box' is not one of the class's
        gpe_pinn.vary_beta.train_gpe_model(0, [0.0], [0], 3, None, 0, 5, [1, 8, 8, 1], 10, 1e-5, 0.01)
