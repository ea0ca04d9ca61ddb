"""Checkpoint interop with the reference.

The reference pickles ONE dict (refine/harmonic_pinn_simulation.py:901-928):
    {'models_state_dicts': {mode: {gamma: {'state_dict', 'layers', 'hbar', 'm', 'mode', 'gamma'}}},
     'mu_table', 'training_history', 'constant_history', 'epochs_history'}
with `state_dict` keys network.{0,2,4,...}.{weight,bias}.  Such files are untrusted input, so they are read with a
restricted unpickler that admits only the three globals the reference's pickles contain (opcode scan of all 23 files
present under /root/reference: collections.OrderedDict, torch._utils._rebuild_tensor_v2, torch.storage._load_from_bytes)
and routes the storage bytes through torch.load(weights_only=True).
"""
from __future__ import annotations

import io
import os
import pickle
from collections import OrderedDict

import numpy as np
import torch


def _safe_load_from_bytes(b):
    return torch.load(io.BytesIO(b), weights_only=True)


class _RestrictedUnpickler(pickle.Unpickler):
    _ALLOWED = {
        ("collections", "OrderedDict"): OrderedDict,
        ("torch._utils", "_rebuild_tensor_v2"): torch._utils._rebuild_tensor_v2,
        ("torch.storage", "_load_from_bytes"): _safe_load_from_bytes,
    }

    def find_class(self, module, name):
        try:
            return self._ALLOWED[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError(f"global {module}.{name} is not allowed in a GPE checkpoint") from None


def load_results(filename, save_dir="."):
    """-> the raw dict of the reference's save_models() (tensors as torch CPU tensors)."""
    with open(os.path.join(save_dir, filename), "rb") as f:
        return _RestrictedUnpickler(f).load()


def load_models(filename="gpe_models.pkl", save_dir=".", flavor="refine"):
    """load_models of refine/harmonic_pinn_simulation.py:931-960: rebuilds one model per (mode, gamma)."""
    from .surface import refine, notebook
    ns = refine if flavor == "refine" else notebook
    data = load_results(filename, save_dir)
    models_by_mode = {}
    for mode, by_gamma in data["models_state_dicts"].items():
        models_by_mode[mode] = {}
        for gamma, md in by_gamma.items():
            model = ns.GrossPitaevskiiPINN(layers=md["layers"], hbar=md["hbar"], m=md["m"], mode=md["mode"], gamma=md["gamma"])
            model.load_state_dict(md["state_dict"])
            models_by_mode[mode][gamma] = model
    return (models_by_mode, data["mu_table"], data["training_history"], data["constant_history"], data["epochs_history"])


def save_models(models_by_mode, mu_table, training_history, constant_history, epochs_history,
                filename="gpe_models.pkl", save_dir="."):
    """save_models of refine/harmonic_pinn_simulation.py:901-928: same dict layout, loadable by the reference."""
    sd = {}
    for mode in models_by_mode:
        sd[mode] = {}
        for gamma, model in models_by_mode[mode].items():
            sd[mode][gamma] = {"state_dict": model.state_dict(), "layers": model.layers, "hbar": model.hbar, "m": model.m,
                               "mode": model.mode, "gamma": model.gamma}
    data = {"models_state_dicts": sd, "mu_table": mu_table, "training_history": training_history,
            "constant_history": constant_history, "epochs_history": epochs_history}
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, filename)
    with open(path, "wb") as f:
        pickle.dump(data, f)
    return path
