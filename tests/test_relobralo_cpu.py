"""ReLoBRaLo weight update (row f4): the package's host-side balancer against a literal restatement of
src/gross_pitaevskii_2D_ReLoBRaLo.py:296-336 with the same torch RNG seed."""
import numpy as np
import torch

from gpe_pinn.relobralo import ReLoBRaLo


def _reference_lambdas(loss_seq, manual, alpha=0.999, temperature=1.0, rho_p=0.9999):
    lambdas, last, init, out = None, None, None, []
    for call_count, losses in enumerate(loss_seq):
        if call_count == 0:
            lambdas = [1.0] * len(losses); last = list(losses); init = list(losses)
        lh = [losses[i] / (last[i] * temperature + 1e-8) for i in range(len(losses))]
        lh = torch.softmax(torch.tensor(lh) - max(lh), dim=0).tolist()
        ih = [losses[i] / (init[i] * temperature + 1e-8) for i in range(len(losses))]
        ih = torch.softmax(torch.tensor(ih) - max(ih), dim=0).tolist()
        rho = torch.bernoulli(torch.tensor(rho_p))
        a = alpha if call_count > 1 else (0.0 if call_count == 1 else 1.0)
        lambdas = [float(rho * a * lambdas[i] + (1 - rho) * a * ih[i] + (1 - a) * lh[i]) for i in range(len(losses))]
        last = list(losses)
        out.append([l * w for l, w in zip(lambdas, manual)])
    return out


def test_update_rule_matches_reference_formula():
    rng = np.random.default_rng(0)
    seq = [list(np.abs(rng.normal(1.0, 0.5, 5)) * np.array([1e-2, 3.0, 50.0, 1e-3, 0.2]) * (0.97 ** k)) for k in range(40)]
    manual = [500.0, 1.0, 2.0, 100.0, 500.0]
    torch.manual_seed(123)
    ref = _reference_lambdas(seq, manual, rho_p=0.7)
    torch.manual_seed(123)
    bal = ReLoBRaLo(5, manual, rho=0.7)
    got = [bal.update(l) for l in seq]
    np.testing.assert_allclose(np.array(got), np.array(ref), rtol=2e-6, atol=1e-9)
    assert abs(sum(got[0]) - sum(manual)) < 1e-9            # first call: all lambdas = 1
