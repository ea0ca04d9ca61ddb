"""Checkpoint interop: the reference's pickle layout written/read by the package, restricted unpickling."""
import os
import pickle

import numpy as np
import pytest
import torch

import gpe_pinn
from gpe_pinn import checkpoint, refine


def test_save_load_roundtrip_reference_layout(tmp_path):
    torch.manual_seed(3)
    m0 = refine.GrossPitaevskiiPINN([1, 64, 64, 64, 1], mode=0, gamma=0.0)
    m1 = refine.GrossPitaevskiiPINN([1, 64, 64, 64, 1], mode=1, gamma=0.5)
    models = {0: {0.0: m0}, 1: {0.5: m1}}
    path = checkpoint.save_models(models, {0: [(0.0, 1.0)], 1: [(0.5, 3.1)]}, {0: {}, 1: {}}, {0: torch.tensor(0.4)},
                                  {0: {0.0: 5001}}, "ck.pkl", str(tmp_path))
    raw = checkpoint.load_results("ck.pkl", str(tmp_path))
    assert set(raw) == {"models_state_dicts", "mu_table", "training_history", "constant_history", "epochs_history"}
    md = raw["models_state_dicts"][1][0.5]
    assert set(md) == {"state_dict", "layers", "hbar", "m", "mode", "gamma"} and md["layers"] == [1, 64, 64, 64, 1]
    assert list(md["state_dict"]) == ["network.0.weight", "network.0.bias", "network.2.weight", "network.2.bias",
                                      "network.4.weight", "network.4.bias", "network.6.weight", "network.6.bias"]
    assert md["state_dict"]["network.2.weight"].shape == (64, 64)
    models2, mu, th, ch, eh = checkpoint.load_models("ck.pkl", str(tmp_path))
    for k, v in m1.state_dict().items():
        assert torch.equal(v, models2[1][0.5].state_dict()[k])
    assert mu[1] == [(0.5, 3.1)] and eh[0][0.0] == 5001
    # a stock torch module (what the reference builds) accepts the state_dict
    net = torch.nn.Sequential(torch.nn.Linear(1, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                              torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 1))
    net.load_state_dict({k.replace("network.", ""): v for k, v in m0.state_dict().items()})


def test_restricted_unpickler_rejects_foreign_globals(tmp_path):
    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    p = tmp_path / "evil.pkl"
    p.write_bytes(pickle.dumps({"models_state_dicts": Evil()}))
    with pytest.raises(pickle.UnpicklingError):
        checkpoint.load_results("evil.pkl", str(tmp_path))


def test_state_dict_shape_mismatch_raises():
    m = refine.GrossPitaevskiiPINN([1, 32, 32, 1])
    sd = m.state_dict()
    sd["network.2.weight"] = torch.zeros(31, 32)
    with pytest.raises(RuntimeError):
        m.load_state_dict(sd)
