"""
oracle/torch_ref.py -- CPU ORACLE (torch-autograd form).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

A restatement, in the reference's own op sequence, of the epoch body of
  /root/reference/Gross-Pitaevskii/src/final/refine/harmonic_pinn_simulation.py:328-361
  /root/reference/Gross_Pitaevskii_1D_power_Test.ipynb c10:L84-103 (classes: c6)
i.e. nn.Sequential forward, torch.autograd.grad(create_graph=True) twice per coordinate
(2D template: src/gross_pitaevskii_2D.py:183-188 with the [N]x[N,1] broadcasting bug Q1 fixed by
slicing [:, k:k+1] as Notebooks/Old/Gross_Pitaevskii/2D_GPE_Riesz_Method_PyTorch.ipynb c0:L97-100 does),
Rayleigh quotient, residual MSE, 10*bc + 20*norm (+5*sym), backward, clip_grad_norm_, Adam, scheduler.

Two uses: (1) it cross-checks the hand-derived reverse pass of oracle/gpe_oracle.py against
autograd, including for the configs the reference does not contain (2D/3D, complex rotating psi);
(2) it is the cost model of "the reference CPU/notebook path" timed by bench.py as cpu_baseline
(kind "port": the reference's Python cannot travel to the GPU box).
It is pinned against the imported reference by tests/golden (see tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from . import gpe_oracle as go


class ShiftedTanh(nn.Module):
    def forward(self, x):                      # harmonic_pinn_simulation.py:48-49
        return torch.tanh(x) + 1.0 + float(np.finfo(float).eps)


class ResidualBlock(nn.Module):                # box_to_gaussian_pinn_simulation.py:52-62
    def __init__(self, dim, dtype):
        super().__init__()
        self.lin1 = nn.Linear(dim, dim, dtype=dtype)
        self.lin2 = nn.Linear(dim, dim, dtype=dtype)

    def forward(self, x):
        return torch.tanh(self.lin2(torch.tanh(self.lin1(x))) + x)


def build_network(layers, activation: int, dtype=torch.float32, net_kind: int = 0) -> nn.Sequential:
    if net_kind == go.NET_RESIDUAL:            # box_to_gaussian_pinn_simulation.py:112-130
        mods = [nn.Linear(layers[0], layers[1], dtype=dtype), ShiftedTanh() if activation == 1 else nn.Tanh()]
        for _ in range(len(layers) - 3):
            mods.append(ResidualBlock(layers[1], dtype))
        mods.append(nn.Linear(layers[1], layers[-1], dtype=dtype))
        return nn.Sequential(*mods)
    mods = []
    for i in range(len(layers) - 1):           # harmonic_pinn_simulation.py:84-93
        mods.append(nn.Linear(layers[i], layers[i + 1], dtype=dtype))
        if i < len(layers) - 2:
            mods.append(ShiftedTanh() if activation == 1 else nn.Tanh())
    return nn.Sequential(*mods)


def set_flat(net: nn.Sequential, flat: np.ndarray):
    o = 0
    with torch.no_grad():
        for p in net.parameters():
            n = p.numel()
            p.copy_(torch.as_tensor(flat[o:o + n], dtype=p.dtype).reshape(p.shape))
            o += n


def get_flat(net: nn.Sequential) -> np.ndarray:
    return np.concatenate([p.detach().cpu().numpy().ravel() for p in net.parameters()])


def get_flat_grad(net: nn.Sequential) -> np.ndarray:
    return np.concatenate([p.grad.detach().cpu().numpy().ravel() for p in net.parameters()])


def weighted_hermite(x: torch.Tensor, n: int) -> torch.Tensor:
    """harmonic_pinn_simulation.py:95-119 (torch recurrence, differentiable)."""
    norm = ((2.0 ** n) * float(math.factorial(n)) * math.sqrt(math.pi)) ** (-0.5)
    if n == 0:
        Hn = torch.ones_like(x)
    elif n == 1:
        Hn = 2.0 * x
    else:
        Hm2, Hm1 = torch.ones_like(x), 2.0 * x
        for k in range(1, n):
            Hn = 2.0 * x * Hm1 - 2.0 * float(k) * Hm2
            Hm2, Hm1 = Hm1, Hn
    return torch.tensor(norm, dtype=x.dtype) * (Hn * torch.exp(-0.5 * x ** 2))


def _potential(pb: go.Problem, x: torch.Tensor, V_pre):
    if pb.potential == go.POT_PRECOMPUTED:
        return V_pre
    if pb.potential == go.POT_HARMONIC:
        V = 0
        for k in range(x.shape[1]):
            V = V + (pb.omega[k] * (x[:, k:k + 1] - (pb.pot_a if k == 0 else 0.0))) ** 2
        return pb.pot_scale * V
    if pb.potential == go.POT_GAUSSIAN:
        return torch.exp(-(x[:, 0:1] - pb.pot_a) ** 2)
    if pb.potential == go.POT_PERIODIC:
        return pb.pot_v0 * torch.cos(pb.pot_k * x[:, 0:1]) ** 2
    if pb.potential == go.POT_NONE:
        return torch.zeros_like(x[:, 0:1])
    raise ValueError(f"Unknown potential type: {pb.potential}")


def epoch_losses(pb: go.Problem, net: nn.Sequential, X: torch.Tensor, x_bc=None, bc_target=None,
                 V_pre=None, detach_lambda: bool = False, orth=None):
    """X requires grad.  Returns (total, dict of pieces).  Mirrors :332-355 / nb c10:L88-97."""
    N, d = X.shape
    u_pred = net(X)                                          # :332
    pert = pb.perturb_scale * u_pred                         # :336-340
    if pb.base_mode >= 0:
        u = weighted_hermite(X, pb.base_mode) + pert         # get_complete_solution :127-134
    else:
        u = pert
    n_out = u.shape[1]
    comps_H = []
    V = _potential(pb, X, V_pre)
    grads1 = []
    laps = []
    for o in range(n_out):
        uo = u[:, o:o + 1]
        g1 = torch.autograd.grad(uo, X, torch.ones_like(uo), create_graph=True, retain_graph=True)[0]   # :158-164
        lap = 0
        for k in range(d):
            gk = g1[:, k:k + 1]
            g2 = torch.autograd.grad(gk, X, torch.ones_like(gk), create_graph=True, retain_graph=True)[0]  # :166-172
            lap = lap + g2[:, k:k + 1]
        grads1.append(g1)
        laps.append(lap)
    if not pb.complex_psi:
        uo = u
        if pb.abs_power:
            inter = pb.gamma * torch.abs(uo) ** (pb.p - 1) * uo
        else:
            inter = pb.gamma * uo ** pb.p                    # :184
        Hu = -pb.kinetic_coeff * laps[0] + V * uo + inter    # :181-186
    else:
        rho = (u * u).sum(dim=1, keepdim=True)
        Hr = -pb.kinetic_coeff * laps[0] + V * u[:, 0:1] + pb.gamma * rho * u[:, 0:1]
        Hi = -pb.kinetic_coeff * laps[1] + V * u[:, 1:2] + pb.gamma * rho * u[:, 1:2]
        if pb.omega_rot != 0.0:
            xx, yy = X[:, 0:1], X[:, 1:2]
            Dr = xx * grads1[0][:, 1:2] - yy * grads1[0][:, 0:1]
            Di = xx * grads1[1][:, 1:2] - yy * grads1[1][:, 0:1]
            Hr = Hr - pb.omega_rot * Di
            Hi = Hi + pb.omega_rot * Dr
        Hu = torch.cat([Hr, Hi], dim=1)
    num = torch.mean((u * Hu).sum(dim=1, keepdim=True))      # :186
    den = torch.mean((u * u).sum(dim=1, keepdim=True))       # :187
    lam = num / den
    if getattr(pb, 'lambda_kind', go.LAMBDA_RAYLEIGH) == go.LAMBDA_ENERGY:
        # src/gross_pitaevskii_2D.py:192  lambda_pde = mean(u_x^2 + u_y^2 + V u^2 + g u^4) / mean(u^2), read per point (quirk Q1: the
        # reference's own tensors broadcast to [N,N]; with ONE point per call the two readings coincide -- tests/golden/make_golden_2d.py)
        e_dens = pb.kinetic_coeff * (grads1[0] ** 2).sum(dim=1, keepdim=True) + V * u ** 2 + pb.gamma * u ** (pb.p + 1)
        lam = torch.mean(e_dens) / torch.mean(u ** 2)
    if detach_lambda:
        lam = lam.detach()
    r = Hu - lam * u                                         # :191
    pde = torch.mean((r * r).sum(dim=1, keepdim=True))       # :194
    integral = torch.sum(u ** 2) * pb.dx                     # :216
    norm = (integral - 1.0) ** 2
    total = pb.w_pde * pde + pb.w_norm * norm
    pieces = dict(pde=pde, norm=norm, lam=lam, u=u, r=r, Hu=Hu, nn=u_pred)
    reg = 0.0
    if getattr(pb, 'w_reg_f', 0.0) != 0.0:                   # src/gross_pitaevskii_2D.py:201  L_f = 1 / (mean(u^2) + 1e-2)
        reg = reg + pb.w_reg_f / (torch.mean(u ** 2) + pb.reg_f_eps)
    if getattr(pb, 'w_reg_lam', 0.0) != 0.0:                 # :204  L_lambda = 1 / (lambda^2 + 1e-6)
        reg = reg + pb.w_reg_lam / (lam ** 2 + pb.reg_lam_eps)
    if not isinstance(reg, float):
        total = total + reg
        pieces['reg'] = reg
    if getattr(pb, 'w_riesz', 0.0) != 0.0:
        kind = getattr(pb, 'riesz_kind', go.RIESZ_PAPER)
        if kind == go.RIESZ_PAPER and d == 1:                # Notebooks/Paper/Gross_Pitaevskii_1D_Harmonic.ipynb c6:L157-177, literally
            dxr = X[1] - X[0]
            norm_factor = torch.sum(u ** 2) * dxr
            kinetic_term = 0.5 * torch.sum(grads1[0] ** 2) * dxr / norm_factor
            potential_term = torch.sum(V * u ** 2) * dxr / norm_factor
            interaction_term = 0.5 * (2.0 * pb.gamma / (pb.p + 1)) * torch.sum(torch.abs(u) ** (pb.p + 1)) * dxr / norm_factor
            riesz = kinetic_term + potential_term + interaction_term
        elif kind == go.RIESZ_SUM and pb.p == 3:             # src/gross_pitaevskii_2D.py:143-149, literally
            laplacian_term = torch.sum(grads1[0] ** 2)
            potential_term = torch.sum(V * u ** 2)
            interaction_term = 0.5 * pb.gamma * torch.sum(u ** 4)
            riesz = 0.5 * (laplacian_term + potential_term + interaction_term)
        else:
            ak, ap, ai, nrm = go.riesz_coefs(pb)
            S = torch.sum(u ** 2)
            if pb.complex_psi:                               # |psi|^(p+1) = rho^((p+1)/2)
                inter = ai * torch.sum((u * u).sum(dim=1) ** (0.5 * (pb.p + 1)))
            else:
                inter = ai * torch.sum(torch.abs(u) ** (pb.p + 1))
            if kind == go.RIESZ_VARIATIONAL:                 # energy of the normalised state u / sqrt(dx sum u^2)
                inter = inter * (pb.dx * S) ** (-0.5 * (pb.p - 1))
            riesz = ak * sum(torch.sum(g1 ** 2) for g1 in grads1) + ap * torch.sum(V * u ** 2) + inter
            if pb.complex_psi and pb.omega_rot != 0.0:       # rotating frame: - Omega <L_z>, L_z = -i (x d_y - y d_x)
                xx, yy = X[:, 0:1], X[:, 1:2]
                Dr = xx * grads1[0][:, 1:2] - yy * grads1[0][:, 0:1]
                Di = xx * grads1[1][:, 1:2] - yy * grads1[1][:, 0:1]
                riesz = riesz - pb.omega_rot * torch.sum(u[:, 0:1] * Di - u[:, 1:2] * Dr)
            if nrm:
                riesz = riesz / S
        total = total + pb.w_riesz * riesz
        pieces['riesz'] = riesz
    if orth is not None and pb.w_orth != 0.0:                # orthogonality penalty (north star; no reference code):
        orth_l = 0                                           #   L_orth = sum_j (dx * sum_m psi_j(x_m) u(x_m))^2
        for j in range(orth.shape[0]):
            orth_l = orth_l + (pb.dx * torch.sum(orth[j].reshape(-1, 1) * u[:, 0:1])) ** 2
        total = total + pb.w_orth * orth_l
        pieces['orth'] = orth_l
    if x_bc is not None and pb.w_bc != 0.0:
        ub = pb.bc_nn_scale * net(x_bc)                      # :202 (quirk Q7: unscaled)
        if pb.base_mode >= 0:
            ub = weighted_hermite(x_bc, pb.base_mode) + ub
        tgt = torch.zeros_like(ub) if bc_target is None else bc_target
        bc = torch.mean((ub - tgt) ** 2)                     # :210
        total = total + pb.w_bc * bc
        pieces['bc'] = bc
    if pb.w_sym != 0.0:                                      # notebook c6:L137-155
        uo, ur = net(X), net(-X)
        sym = torch.mean((uo - pb.sym_sign * ur) ** 2)
        total = total + pb.w_sym * sym
        pieces['sym'] = sym
    return total, pieces


class TorchTrainer:
    """Adam + clip + scheduler exactly as the reference drives them."""

    def __init__(self, pb: go.Problem, flat: np.ndarray, x: np.ndarray, x_bc=None, lr=1e-3,
                 sched=go.SCHED_CONST, dtype=torch.float32, clip_norm=1.0):
        self.pb = pb
        self.net = build_network(list(pb.layers), pb.activation, dtype, getattr(pb, "net_kind", 0))
        set_flat(self.net, flat)
        self.X = torch.tensor(x, dtype=dtype, requires_grad=True)
        self.x_bc = None if x_bc is None else torch.tensor(x_bc, dtype=dtype)
        self.opt = torch.optim.Adam(self.net.parameters(), lr=lr)            # :309
        self.clip_norm = clip_norm
        self.sched_kind = sched
        if sched == go.SCHED_COSINE_LOSS:
            self.sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(
                self.opt, T_0=200, T_mult=2, eta_min=1e-6)                   # :312-314
        elif sched == go.SCHED_PLATEAU:
            self.sched = torch.optim.lr_scheduler.ReduceLROnPlateau(
                self.opt, mode='min', factor=0.5, patience=100, min_lr=1e-5)  # nb c10:L76-78
        else:
            self.sched = None

    def step(self):
        self.opt.zero_grad()                                                 # :329
        total, pieces = epoch_losses(self.pb, self.net, self.X, self.x_bc)
        total.backward()                                                     # :358
        gn = 0.0
        if self.clip_norm > 0:
            gn = float(torch.nn.utils.clip_grad_norm_(self.net.parameters(), self.clip_norm))  # :359
        lr_used = self.opt.param_groups[0]['lr']
        self.opt.step()                                                      # :360
        if self.sched is not None:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self.sched.step(total)                                       # :361 (quirk Q4) / nb c10:L103
        return dict(loss=float(total), pde=float(pieces['pde']), norm=float(pieces['norm']),
                    bc=float(pieces.get('bc', 0.0)), sym=float(pieces.get('sym', 0.0)),
                    mu=float(pieces['lam']), grad_norm=gn, lr=lr_used)

    def flat(self):
        return get_flat(self.net)
