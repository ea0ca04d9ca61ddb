"""CPU: oracle/cpu_ref (C++/OpenMP jet restatement, fp32) against the numpy oracle, and its AddressSanitizer + UBSan build."""
import numpy as np
import pytest

from oracle import cpu_ref
from oracle import gpe_oracle as go

CASES = {
    "1d_64x4_g100": (dict(layers=[1, 64, 64, 64, 64, 1], gamma=100.0, dx=20 / 4095, w_bc=0.0), 4096),
    "2d_64x4_g500": (dict(layers=[2, 64, 64, 64, 64, 1], gamma=500.0, dx=0.004, w_bc=0.0), 2000),
    "3d_32x3_aniso": (dict(layers=[3, 32, 32, 32, 1], gamma=50.0, dx=0.002, omega=(1.0, 1.4, 2.0), w_bc=0.0), 777),
    "1d_shifted_tanh_p4": (dict(layers=[1, 32, 32, 1], activation=1, gamma=2.0, p=4, kinetic_coeff=1.0, pot_scale=1.0, dx=0.01, w_bc=0.0), 33),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("threads", [1, 4])
def test_cpu_ref_matches_numpy_oracle(name, threads):
    kw, N = CASES[name]
    rng = np.random.default_rng(4)
    d = kw["layers"][0]
    x = rng.uniform(-3, 3, (N, d)).astype(np.float32)
    flat = (rng.normal(0, 1, go.param_count(kw["layers"])) * 0.3).astype(np.float32)
    pb = go.Problem(**kw)
    osc, ograd, _ = go.full_loss_and_grad(pb, flat.astype(np.float64), x.astype(np.float64))
    sc, g = cpu_ref.loss_grad(cpu_ref.load(), pb, flat, x, threads=threads)
    for k, tol in (("mu", 2e-5), ("loss", 2e-4), ("pde", 2e-4), ("norm", 2e-4)):
        assert abs(sc[k] - osc[k]) <= tol * max(abs(osc[k]), 1e-6), (k, sc[k], osc[k])
    assert np.abs(g - ograd).max() <= 2e-4 * np.abs(ograd).max()


def test_cpu_ref_rejects_what_it_does_not_model():
    with pytest.raises(ValueError):
        cpu_ref.loss_grad(cpu_ref.load(), go.Problem(layers=[2, 8, 8, 2], complex_psi=True), np.zeros(go.param_count([2, 8, 8, 2])),
                          np.zeros((4, 2)))


def test_address_and_ub_sanitizer_build_runs_clean():
    """sanitizers run on the CPU restatement (GPU AddressSanitizer / XNACK runs are not available on this pool)."""
    out = cpu_ref.sanitizer_selftest()
    assert "0 finite-difference mismatches" in out
