"""Host-side ReLoBRaLo loss balancing (row f4 of SURVEY 8) for the engine's per-term scalars.

The update rule is the one of src/gross_pitaevskii_2D_ReLoBRaLo.py:296-336 (alpha = 0.999, temperature = 1, rho = 0.9999,
softmax of loss ratios against the previous and the initial losses, Bernoulli(rho) look-back, manual weights on top).
There the balancing weights depend on the CURRENT losses and multiply them in the same backward pass, so a step is:
    losses = engine.residual()  ->  weights = balancer.update(losses)  ->  engine.set_loss_weights(...)  ->  engine.step()
(one extra forward per step and one host synchronisation, as the reference's .item() calls imply).
"""
from __future__ import annotations

import math
from typing import Sequence

import torch


class ReLoBRaLo:
    def __init__(self, n_terms: int, manual_weights: Sequence[float] = None, alpha: float = 0.999, temperature: float = 1.0,
                 rho: float = 0.9999):
        self.n = int(n_terms)
        self.manual = list(manual_weights) if manual_weights is not None else [1.0] * self.n
        self.alpha, self.temperature, self.rho = float(alpha), float(temperature), float(rho)
        self.call_count = 0
        self.lambdas = [1.0] * self.n
        self.last_losses = None
        self.init_losses = None

    @staticmethod
    def _softmax(v):
        m = max(v)
        e = [math.exp(x - m) for x in v]
        s = sum(e)
        return [x / s for x in e]

    def update(self, losses: Sequence[float]):
        """losses: current value of every term.  Returns the weights (lambda_i * manual_i) to use for THIS step."""
        losses = [float(x) for x in losses]
        assert len(losses) == self.n
        if self.call_count == 0:                                                   # :301-305
            self.lambdas = [1.0] * self.n
            self.last_losses = list(losses)
            self.init_losses = list(losses)
        lam_hat = self._softmax([losses[i] / (self.last_losses[i] * self.temperature + 1e-8) for i in range(self.n)])
        init_hat = self._softmax([losses[i] / (self.init_losses[i] * self.temperature + 1e-8) for i in range(self.n)])
        rho = float(torch.bernoulli(torch.tensor(self.rho)))                       # :320 (same RNG stream as the reference)
        alpha = self.alpha if self.call_count > 1 else (0.0 if self.call_count == 1 else 1.0)      # :321
        self.lambdas = [rho * alpha * self.lambdas[i] + (1 - rho) * alpha * init_hat[i] + (1 - alpha) * lam_hat[i]
                        for i in range(self.n)]
        self.last_losses = list(losses)
        self.call_count += 1
        return [l * w for l, w in zip(self.lambdas, self.manual)]


def balanced_step(engine, balancer: ReLoBRaLo, terms=("bc", "riesz", "pde", "norm", "sym")):
    """One ReLoBRaLo-balanced training step on the engine (term order of the reference: data, riesz, pde, norm, sym)."""
    sc, _, _ = engine.residual(want_fields=False)
    w = dict(zip(terms, balancer.update([sc[t] for t in terms])))
    engine.set_loss_weights(w.get("pde", 0.0), w.get("bc", 0.0), w.get("norm", 0.0), w.get("sym", 0.0), w.get("orth", 0.0),
                            w.get("riesz", 0.0))
    return engine.step(), w
