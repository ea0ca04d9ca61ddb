"""
oracle/gpe_oracle.py -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (the package + libgpe_hip.so) never does.

What it is: a numpy restatement of the Gross-Pitaevskii eigenvalue-residual training
step of the reference, written as forward-mode derivative "jets" (value, d first
derivatives, d diagonal second derivatives) through the MLP plus a hand-derived
reverse pass, so that it runs in fp64 or fp32 with no autograd.  It is pinned by
tests/golden/*.npz, which were produced HERE by importing the reference itself
(tests/golden/make_golden.py); see tests/test_oracle_golden.py.

Reference anchors (under /root/reference/Gross-Pitaevskii/src/final/refine/ unless noted):
  MLP + activation           harmonic_pinn_simulation.py:41-49, 84-93, 121-125
  Hermite base               harmonic_pinn_simulation.py:95-119 (notebook c6:L41-48)
  potential                  harmonic_pinn_simulation.py:136-144 (notebook c6:L63-79)
  d/dx, d2/dx2 (autograd)    harmonic_pinn_simulation.py:158-172 ; 2D: src/gross_pitaevskii_2D.py:183-188
  Rayleigh quotient, residual harmonic_pinn_simulation.py:181-194 (notebook c6:L113-125)
  boundary / normalisation   harmonic_pinn_simulation.py:198-217
  symmetry                   Gross_Pitaevskii_1D_power_Test.ipynb c6:L137-155
  epoch body                 harmonic_pinn_simulation.py:328-361 ; notebook c10:L84-103
  clip / Adam / schedulers   harmonic_pinn_simulation.py:309-314, 359-361 ; notebook c10:L73-78,L101-103
Configs 3-5 of BASELINE.json (2D/3D harmonic, complex rotating psi, orthogonality) have no
reference counterpart: for those this file is the definition ("parity unpinned" by the
reference; pinned by the torch-autograd restatement oracle/torch_ref.py in fp64 --
tests/test_oracle_autograd.py -- and, for the 2D Laplacian / residual of one point, by the
reference's own 2D class called one point at a time, where its broadcast quirk Q1 is inert:
tests/golden/fx_2d_ref_points.npz, tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

# ----------------------------------------------------------------------------------------
# problem description (oracle-side twin of include/gpe_hip.h:gpe_config)
# ----------------------------------------------------------------------------------------
POT_HARMONIC, POT_GAUSSIAN, POT_PERIODIC, POT_PRECOMPUTED, POT_NONE = 0, 1, 2, 3, 4
SCHED_CONST, SCHED_COSINE_LOSS, SCHED_PLATEAU = 0, 1, 2
BASE_HERMITE, BASE_BOX, BASE_PRECOMPUTED = 0, 1, 2
ENV_NONE, ENV_SIN = 0, 1
RIESZ_PAPER, RIESZ_SUM, RIESZ_VARIATIONAL = 0, 1, 2
LAMBDA_RAYLEIGH, LAMBDA_ENERGY = 0, 1      # eigenvalue estimate inside the residual: sum u Hu / sum u^2 | src/gross_pitaevskii_2D.py:192
NET_MLP, NET_RESIDUAL = 0, 1


@dataclass
class Problem:
    layers: Sequence[int]                      # [d, H, ..., out]
    activation: int = 0                        # 0 tanh, 1 tanh+1 (ShiftedTanh)
    complex_psi: bool = False                  # out must be 2: (Re, Im)
    kinetic_coeff: float = 0.5                 # c in  -c * laplacian
    potential: int = POT_HARMONIC
    pot_scale: float = 0.5                     # harmonic: V = pot_scale * sum (omega_k x_k)^2
    omega: Sequence[float] = (1.0, 1.0, 1.0)
    pot_a: float = 0.0                         # gaussian centre; harmonic: centre of the trap along x
    pot_v0: float = 1.0                        # periodic depth
    pot_k: float = 2 * math.pi / 5.0           # periodic wave number
    omega_rot: float = 0.0                     # rotation frequency (complex psi, d>=2)
    gamma: float = 0.0
    p: int = 3                                 # nonlinearity power:  gamma * u**p
    abs_power: bool = False                    # gamma * |u|^(p-1) * u
    base_mode: int = -1                        # -1: no analytic base; n>=0: 1D Hermite phi_n
    base_deriv: int = 0                        # 0 exact derivative, 1 notebook quirk (H_n constant)
    perturb_scale: float = 1.0                 # multiplies the NN output in pde/norm terms
    bc_nn_scale: float = 1.0                   # multiplies the NN output in the boundary term (quirk Q7)
    w_pde: float = 1.0
    w_bc: float = 10.0
    w_norm: float = 20.0
    w_sym: float = 0.0
    w_orth: float = 0.0
    w_riesz: float = 0.0                       # Riesz energy term (real psi); see riesz_coefs()
    riesz_kind: int = RIESZ_PAPER              # PAPER: Paper nb c6:L133-183 ; SUM: src/gross_pitaevskii_2D.py:112-151 ; VARIATIONAL
    sym_sign: float = 1.0                      # +1 even mode, -1 odd mode
    base_kind: int = BASE_HERMITE              # refine/box_pinn_simulation.py:99-117 (BOX); caller arrays (PRECOMPUTED)
    envelope: int = ENV_NONE                   # ENV_SIN: forward = NN * sin(pi x / env_L)  (refine/box_pinn_simulation.py:119-130)
    box_L: float = 1.0
    env_L: float = 1.0
    net_kind: int = NET_MLP                    # NET_RESIDUAL: refine/box_to_gaussian_pinn_simulation.py:52-63,100-130 (layers = [d, H, ..., H, out],
                                               #   len(layers) - 3 residual blocks tanh(lin2(tanh(lin1 x)) + x) behind Linear(d,H) + activation)
    lambda_kind: int = LAMBDA_RAYLEIGH         # LAMBDA_ENERGY: lambda = [c sum |grad u|^2 + sum V u^2 + gamma sum |u|^(p+1)] / sum u^2  (2D classes)
    w_reg_f: float = 0.0                       # w / (mean u^2 + reg_f_eps)      src/gross_pitaevskii_2D.py:201
    reg_f_eps: float = 1e-2
    w_reg_lam: float = 0.0                     # w / (lambda^2 + reg_lam_eps)    src/gross_pitaevskii_2D.py:204
    reg_lam_eps: float = 1e-6
    dx: float = 1.0                            # quadrature weight
    n_global: int = 0                          # N used in the means (0 -> len(x))

    @property
    def dim(self) -> int:
        return int(self.layers[0])

    @property
    def n_out(self) -> int:
        return int(self.layers[-1])


def expand_layers(layers: Sequence[int], net_kind: int = NET_MLP):
    """-> (widths of the linear maps' ends, skip[j] = hidden layer whose jets are added to map j's output before its activation or -1,
    plain_tanh[h] = True when hidden layer h uses tanh without the ShiftedTanh shift)."""
    layers = [int(v) for v in layers]
    if net_kind == NET_MLP:
        n_lin = len(layers) - 1
        return layers, [-1] * n_lin, [False] * (n_lin - 1)
    assert len(layers) >= 4 and len(set(layers[1:-1])) == 1, "residual network: layers = [d, H, ..., H, out]"
    nb, H = len(layers) - 3, layers[1]
    widths = [layers[0]] + [H] * (2 * nb + 1) + [layers[-1]]
    skip = [-1] * (2 * nb + 2)
    for b in range(nb):
        skip[2 * b + 2] = 2 * b                 # lin2 of block b adds the block input = hidden layer 2b
    return widths, skip, [False] + [True] * (2 * nb)


def param_count(layers: Sequence[int], net_kind: int = NET_MLP) -> int:
    w = expand_layers(layers, net_kind)[0]
    return sum(w[i] * w[i + 1] + w[i + 1] for i in range(len(w) - 1))


def unflatten(flat: np.ndarray, layers: Sequence[int], net_kind: int = NET_MLP) -> List[Tuple[np.ndarray, np.ndarray]]:
    """flat is in torch state_dict order: network.0.weight [out,in], network.0.bias, network.2.weight ...
    (residual network: network.0, network.2.lin1, network.2.lin2, network.3.lin1, ..., output Linear)"""
    w = expand_layers(layers, net_kind)[0]
    out, o = [], 0
    for i in range(len(w) - 1):
        fi, fo = w[i], w[i + 1]
        W = flat[o:o + fi * fo].reshape(fo, fi); o += fi * fo
        b = flat[o:o + fo]; o += fo
        out.append((W, b))
    assert o == flat.size
    return out


def flatten(params: List[Tuple[np.ndarray, np.ndarray]]) -> np.ndarray:
    return np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in params])


# ----------------------------------------------------------------------------------------
# MLP with derivative jets.  Channel layout: 0 = value, 1..d = d/dx_k, d+1..2d = d2/dx_k^2
# (value_only -> a single channel).
# ----------------------------------------------------------------------------------------
def mlp_forward(params, x: np.ndarray, activation: int, value_only: bool = False, skip=None, plain_tanh=None):
    """x [N,d] -> out jets [C,N,n_out]; cache for the reverse pass.  skip / plain_tanh: expand_layers() of a residual network."""
    dt = x.dtype
    N, d = x.shape
    C = 1 if value_only else 1 + 2 * d
    A = np.zeros((C, N, d), dtype=dt)
    A[0] = x
    if not value_only:
        for k in range(d):
            A[1 + k, :, k] = 1
    cache = []
    acts = []
    L = len(params)
    for l, (W, b) in enumerate(params):
        Z = A @ W.T.astype(dt)                  # [C,N,out]   (harmonic_pinn_simulation.py:90 nn.Linear)
        Z[0] += b.astype(dt)
        if l == L - 1:
            cache.append((A, None, None, Z))
            return Z, cache
        if skip is not None and skip[l] >= 0:   # box_to_gaussian_pinn_simulation.py:58-62: tanh(lin2(...) + identity)
            Z = Z + acts[skip[l]]
        shift = dt.type(1.0) if (activation == 1 and not (plain_tanh is not None and plain_tanh[l])) else dt.type(0.0)
        t = np.tanh(Z[0])                       # harmonic_pinn_simulation.py:48-49 / notebook c6:L38
        s = 1 - t * t
        An = np.empty_like(Z)
        An[0] = t + shift
        if not value_only:
            for k in range(d):
                zk, zkk = Z[1 + k], Z[1 + d + k]
                An[1 + k] = s * zk
                An[1 + d + k] = s * zkk - 2 * t * s * zk * zk
        cache.append((A, t, s, Z))
        acts.append(An)
        A = An


def mlp_backward(params, cache, out_bar: np.ndarray, value_only: bool = False, skip=None):
    """out_bar [C,N,n_out] = dLoss/d(out jets).  Returns the flat gradient (torch state_dict order)."""
    L = len(params)
    grads = [None] * L
    d = params[0][0].shape[1]
    Zb = out_bar
    pending = {}                                 # hidden layer index -> adjoint arriving over a skip connection
    for l in range(L - 1, -1, -1):
        W, b = params[l]
        A, t, s, Z = cache[l]
        dt = A.dtype
        if l < L - 1:
            Ab = Zb                              # adjoint of this layer's activation jets
            if l in pending:
                Ab = Ab + pending.pop(l)
            Zb = np.empty_like(Ab)
            ts2 = -2 * t * s
            acc = s * Ab[0]
            if not value_only:
                q = -2 * s * s + 4 * t * t * s
                for k in range(d):
                    zk, zkk = Z[1 + k], Z[1 + d + k]
                    akb, akkb = Ab[1 + k], Ab[1 + d + k]
                    Zb[1 + d + k] = s * akkb
                    Zb[1 + k] = s * akb - 4 * t * s * zk * akkb
                    acc = acc + ts2 * zk * akb + (ts2 * zkk + q * zk * zk) * akkb
            Zb[0] = acc
            if skip is not None and skip[l] >= 0:
                pending[skip[l]] = Zb            # z = lin(a) + a_skip: the same adjoint flows into the skipped-from activations
        # weight / bias gradients:  dW = sum_c Zb_c^T A_c ; db = sum_m Zb_0
        gW = np.einsum('cmo,cmi->oi', Zb, A)
        gb = Zb[0].sum(axis=0)
        grads[l] = (gW.astype(dt), gb.astype(dt))
        if l > 0:
            Zb = Zb @ W.astype(dt)               # adjoint of the input jets  [C,N,in]
    return flatten(grads)


# ----------------------------------------------------------------------------------------
# analytic base, potential
# ----------------------------------------------------------------------------------------
def hermite_base(x1: np.ndarray, n: int, deriv_mode: int = 0):
    """phi_n(x), phi_n', phi_n'' for the 1D harmonic oscillator
    (harmonic_pinn_simulation.py:95-119; derivatives there come from autograd)."""
    dt = x1.dtype
    if n < 0:
        z = np.zeros_like(x1)
        return z, z.copy(), z.copy()
    norm = dt.type(((2.0 ** n) * float(math.factorial(n)) * math.sqrt(math.pi)) ** (-0.5))
    Hm2 = np.zeros_like(x1)            # H_{n-2}
    Hm1 = np.zeros_like(x1)            # H_{n-1}
    H = np.ones_like(x1)               # H_0
    for k in range(n):                 # H_{k+1} = 2x H_k - 2k H_{k-1}
        Hn = 2 * x1 * H - 2 * dt.type(k) * Hm1
        Hm2, Hm1, H = Hm1, H, Hn
    w = np.exp(dt.type(-0.5) * x1 * x1)
    if deriv_mode == 1:                # notebook c6:L45-47: H_n enters as a constant tensor (quirk Q8)
        H1 = np.zeros_like(x1)
        H2 = np.zeros_like(x1)
    else:
        H1 = 2 * dt.type(n) * Hm1                       # H_n'  = 2n H_{n-1}
        H2 = 4 * dt.type(n) * dt.type(n - 1) * Hm2      # H_n'' = 4n(n-1) H_{n-2}
    phi = norm * (H * w)
    phi1 = norm * w * (H1 - x1 * H)
    phi2 = norm * w * (H2 - 2 * x1 * H1 + (x1 * x1 - 1) * H)
    return phi, phi1, phi2


def base_functions(pb: Problem, x1: np.ndarray, base_pre=None):
    """phi_n, phi_n', phi_n'' of the configured base on 1D points."""
    dt = x1.dtype
    if pb.base_kind == BASE_PRECOMPUTED:
        return tuple(np.asarray(a, dtype=dt) for a in base_pre)
    if pb.base_kind == BASE_BOX:                       # refine/box_pinn_simulation.py:99-117, 141-180
        k = dt.type((pb.base_mode + 1) * math.pi / pb.box_L)
        a = dt.type(math.sqrt(2.0 / pb.box_L))
        return a * np.sin(k * x1), a * k * np.cos(k * x1), -a * k * k * np.sin(k * x1)
    return hermite_base(x1, pb.base_mode, pb.base_deriv)


def envelope(pb: Problem, x1: np.ndarray):
    dt = x1.dtype
    k = dt.type(math.pi / pb.env_L)
    return np.sin(k * x1), k * np.cos(k * x1), -k * k * np.sin(k * x1)


def potential(pb: Problem, x: np.ndarray, V_pre: Optional[np.ndarray] = None) -> np.ndarray:
    dt = x.dtype
    if pb.potential == POT_PRECOMPUTED:
        return V_pre.astype(dt)
    if pb.potential == POT_HARMONIC:            # harmonic_pinn_simulation.py:141 / notebook c6:L69
        V = np.zeros(x.shape[0], dtype=dt)
        for k in range(x.shape[1]):
            w = dt.type(pb.omega[k])
            c = dt.type(pb.pot_a if k == 0 else 0.0)   # trap centre along x (refine/vary_potential_parameter_harmonic.py:231-240)
            V = V + (w * (x[:, k] - c)) ** 2
        return dt.type(pb.pot_scale) * V
    if pb.potential == POT_GAUSSIAN:            # notebook c6:L71-72
        return np.exp(-(x[:, 0] - dt.type(pb.pot_a)) ** 2)
    if pb.potential == POT_PERIODIC:            # notebook c6:L74-76
        return dt.type(pb.pot_v0) * np.cos(dt.type(pb.pot_k) * x[:, 0]) ** 2
    if pb.potential == POT_NONE:
        return np.zeros(x.shape[0], dtype=dt)
    raise ValueError(f"Unknown potential type: {pb.potential}")


def riesz_coefs(pb: "Problem"):
    """E = (ak sum |grad u|^2 + ap sum V u^2 + ai sum |u|^(p+1)) / (normalised ? sum u^2 : 1)  -> (ak, ap, ai, normalised)"""
    gi = pb.gamma / (pb.p + 1)
    if pb.riesz_kind == RIESZ_SUM:             # src/gross_pitaevskii_2D.py:143-149 (p = 3: 1/2 (K + P + 1/2 eta sum u^4))
        return 0.5, 0.5, gi, False
    if pb.riesz_kind == RIESZ_VARIATIONAL:     # energy of the normalised state: the interaction sum also carries I^(-(p-1)/2), I = dx sum u^2
        return pb.kinetic_coeff, 1.0, 2.0 * gi, True
    return 0.5, 1.0, gi, True                  # Paper nb c6:L163-177


def energy_numerator(pb: "Problem", tot: dict) -> float:
    """c sum |grad u|^2 + sum V u^2 + gamma sum |u|^(p+1) from the three energy sums (filed with the Riesz coefficients on them)."""
    ak, ap, _, _ = riesz_coefs(pb)
    ci = 0.5 * (pb.p + 1) if pb.riesz_kind == RIESZ_VARIATIONAL else float(pb.p + 1)
    return pb.kinetic_coeff * tot['rz_k'] / ak + tot['rz_p'] / ap + ci * tot['rz_i']


def reg_terms(pb: "Problem", den: float, lam: float, N: int) -> float:
    """src/gross_pitaevskii_2D.py:197-211: L_f = w / (mean u^2 + eps), L_lambda = w / (lambda^2 + eps)"""
    r = 0.0
    if pb.w_reg_f != 0.0:
        r += pb.w_reg_f / (den / N + pb.reg_f_eps)
    if pb.w_reg_lam != 0.0:
        r += pb.w_reg_lam / (lam * lam + pb.reg_lam_eps)
    return r


def _ipow(u, p: int):
    r = np.ones_like(u)
    for _ in range(p):
        r = r * u
    return r


# ----------------------------------------------------------------------------------------
# head: NN output jets -> u, H u, sums
# ----------------------------------------------------------------------------------------
def head_pde(pb: Problem, x: np.ndarray, out: np.ndarray, V_pre=None, base_pre=None):
    """out [C,N,n_out].  Returns dict with u, Hu ([N,n_out]), V, and the jets of u."""
    dt = x.dtype
    N, d = x.shape
    sc = dt.type(pb.perturb_scale)
    if pb.envelope == ENV_SIN:                    # model.forward = network(x) * sin(pi x)  (box_pinn_simulation.py:127-130)
        assert d == 1 and pb.n_out == 1
        f, f1, f2 = (a[:, None] for a in envelope(pb, x[:, 0]))
        o0, o1, o2 = out[0], out[1], out[2]
        out = np.stack([o0 * f, o1 * f + o0 * f1, o2 * f + 2 * o1 * f1 + o0 * f2])
    U = sc * out                                  # harmonic_pinn_simulation.py:336-340
    if pb.base_mode >= 0:                         # get_complete_solution :127-134
        assert d == 1 and pb.n_out == 1
        phi, phi1, phi2 = base_functions(pb, x[:, 0], base_pre)
        U = U.copy()
        U[0, :, 0] += phi
        U[1, :, 0] += phi1
        U[2, :, 0] += phi2
    V = potential(pb, x, V_pre)
    c = dt.type(pb.kinetic_coeff)
    g = dt.type(pb.gamma)
    u = U[0]                                      # [N,n_out]
    lap = U[1 + d:1 + 2 * d].sum(axis=0)          # [N,n_out]
    if not pb.complex_psi:
        if pb.abs_power:                          # Paper nb c6:L117 form
            inter = g * _ipow(np.abs(u), pb.p - 1) * u
        else:                                     # :184  gamma * u**p
            inter = g * _ipow(u, pb.p)
        Hu = -c * lap + V[:, None] * u + inter    # :181-186
    else:
        assert pb.n_out == 2 and pb.p == 3
        rho = (u * u).sum(axis=1, keepdims=True)
        Hu = -c * lap + V[:, None] * u + g * rho * u
        if pb.omega_rot != 0.0:
            Om = dt.type(pb.omega_rot)
            xx, yy = x[:, 0], x[:, 1]
            Dr = xx * U[2, :, 0] - yy * U[1, :, 0]      # (x d_y - y d_x) psi_r
            Di = xx * U[2, :, 1] - yy * U[1, :, 1]
            Hu = Hu.copy()
            Hu[:, 0] += -Om * Di                        # -Omega L_z psi = i Omega (x d_y - y d_x) psi
            Hu[:, 1] += Om * Dr
    return dict(U=U, u=u, Hu=Hu, V=V)


def loss_and_grad(pb: Problem, flat: np.ndarray, x: np.ndarray, x_bc: Optional[np.ndarray] = None,
                  bc_target: Optional[np.ndarray] = None, V_pre=None,
                  orth: Optional[np.ndarray] = None, want_grad: bool = True,
                  shard_sums: Optional[dict] = None, phase: int = 0, base_pre=None):
    """One evaluation of the epoch body (harmonic_pinn_simulation.py:328-358 / notebook c10:L85-100)
    up to and including backward().  lambda is treated as a constant in the reverse pass
    (its branch is identically zero).

    Data-parallel use: call with phase=1 on a shard to get the local sums, add them over shards,
    pass the totals back as shard_sums with phase=2 to get the shard's gradient / loss pieces.
    """
    dt = x.dtype
    N_loc, d = x.shape
    N = pb.n_global if pb.n_global > 0 else N_loc
    params = unflatten(flat.astype(dt), pb.layers, pb.net_kind)
    _, skip, plain = expand_layers(pb.layers, pb.net_kind)
    out, cache = mlp_forward(params, x, pb.activation, skip=skip, plain_tanh=plain)
    h = head_pde(pb, x, out, V_pre, base_pre)
    u, Hu, V, U = h['u'], h['Hu'], h['V'], h['U']
    acc = np.float64
    sums = dict(num=float((u * Hu).sum(dtype=acc)), den=float((u * u).sum(dtype=acc)))
    n_orth = 0 if orth is None else orth.shape[0]
    for j in range(n_orth):
        sums[f'orth{j}'] = float((orth[j].astype(dt)[:, None] * u).sum(dtype=acc))
    if pb.w_riesz != 0.0 or pb.lambda_kind == LAMBDA_ENERGY:     # Paper nb c6:L163-174 (dx cancels in the quotient) ; 2D: src/...2D.py:112-151,192
        ak, ap, ai, _ = riesz_coefs(pb)
        sums['rz_k'] = float((dt.type(ak) * (U[1:1 + d] ** 2).sum(axis=0)).sum(dtype=acc))
        sums['rz_p'] = float((dt.type(ap) * V[:, None] * u * u).sum(dtype=acc))
        if not pb.complex_psi:
            sums['rz_i'] = float((dt.type(ai) * _ipow(np.abs(u), pb.p + 1)).sum(dtype=acc))
            sums['rz_l'] = 0.0
        else:
            # complex psi (north star; no reference code): |psi|^(p+1) = rho^((p+1)/2), and the rotating-frame term -Omega <L_z>,
            # <L_z> = sum psi_r (x d_y - y d_x) psi_i - psi_i (x d_y - y d_x) psi_r   (quadratic in psi like the kinetic and potential sums)
            assert pb.n_out == 2 and pb.p == 3
            rho = (u * u).sum(axis=1)
            sums['rz_i'] = float((dt.type(ai) * rho * rho).sum(dtype=acc))
            sums['rz_l'] = 0.0
            if pb.omega_rot != 0.0:
                xx, yy = x[:, 0].astype(dt), x[:, 1].astype(dt)
                Dr = xx * U[2, :, 0] - yy * U[1, :, 0]
                Di = xx * U[2, :, 1] - yy * U[1, :, 1]
                sums['rz_l'] = float((u[:, 0] * Di - u[:, 1] * Dr).sum(dtype=acc))
    # symmetry term: two value-only passes (notebook c6:L143-147)
    sym = None
    if pb.w_sym != 0.0:
        o1, c1 = mlp_forward(params, x, pb.activation, value_only=True, skip=skip, plain_tanh=plain)
        o2, c2 = mlp_forward(params, -x, pb.activation, value_only=True, skip=skip, plain_tanh=plain)
        diff = o1[0] - dt.type(pb.sym_sign) * o2[0]
        sums['sym'] = float((diff * diff).sum(dtype=acc))
        sym = (diff, c1, c2)
    if phase == 1:
        return sums
    if shard_sums is not None:
        tot = shard_sums
    else:
        tot = sums
    if pb.lambda_kind == LAMBDA_ENERGY:                          # src/gross_pitaevskii_2D.py:192 (per-point reading of quirk Q1)
        assert not pb.complex_psi and pb.n_out == 1 and pb.p % 2 == 1
        lam = dt.type(energy_numerator(pb, tot) / tot['den'])
    else:
        lam = dt.type(tot['num'] / tot['den'])                   # :186-188
    r = Hu - lam * u                                             # :191
    sr2 = float((r * r).sum(dtype=acc))
    I = dt.type(tot['den']) * dt.type(pb.dx)                     # :216 torch.sum(u**2)*dx
    res = dict(lam=float(lam), sum_r2=sr2, integral=float(I), sums=sums)
    L_norm = float((I - 1) ** 2)
    # boundary term (replicated on every shard; :198-210)
    L_bc = 0.0
    if x_bc is not None and pb.w_bc != 0.0:
        ob, cb = mlp_forward(params, x_bc.astype(dt), pb.activation, value_only=True, skip=skip, plain_tanh=plain)
        fenv_b = envelope(pb, x_bc[:, 0].astype(dt))[0][:, None] if pb.envelope == ENV_SIN else dt.type(1.0)
        fb = dt.type(pb.bc_nn_scale) * fenv_b * ob[0]
        if pb.base_mode >= 0 and pb.base_kind != BASE_PRECOMPUTED:
            fb = fb + base_functions(pb, x_bc[:, 0].astype(dt))[0][:, None]
        tgt = np.zeros_like(fb) if bc_target is None else bc_target.astype(dt).reshape(fb.shape)
        eb = fb - tgt
        L_bc = float((eb * eb).mean(dtype=acc))                  # torch.mean over all elements (:210)
    L_orth = 0.0
    for j in range(n_orth):
        L_orth += (tot[f'orth{j}'] * pb.dx) ** 2
    L_sym = (tot['sym'] / N) if sym is not None else 0.0
    fI = 1.0
    if pb.w_riesz != 0.0:
        rz_norm = riesz_coefs(pb)[3]
        if pb.riesz_kind == RIESZ_VARIATIONAL:
            fI = (tot['den'] * pb.dx) ** (-0.5 * (pb.p - 1))
        E_rz = (tot['rz_k'] + tot['rz_p'] + fI * tot['rz_i'] - pb.omega_rot * tot.get('rz_l', 0.0)) / (tot['den'] if rz_norm else 1.0)
    else:
        E_rz = 0.0
    res.update(L_norm=L_norm, L_bc=L_bc, L_orth=L_orth, L_sym=L_sym, L_riesz=E_rz, L_reg=reg_terms(pb, tot['den'], float(lam), N))
    if not want_grad:
        return res
    # ---- seeds ----
    n_out = pb.n_out
    C = 1 + 2 * d
    rb = (dt.type(2.0 * pb.w_pde / N)) * r                       # d(w_pde*mean(r^2))/dr
    ub = np.zeros_like(u)
    g = dt.type(pb.gamma)
    if not pb.complex_psi:
        if pb.abs_power:
            dinter = g * dt.type(pb.p) * _ipow(np.abs(u), pb.p - 1)
        else:
            dinter = g * dt.type(pb.p) * _ipow(u, pb.p - 1)
        ub = rb * (V[:, None] + dinter - lam)
    else:
        ur, ui = u[:, 0], u[:, 1]
        rho = ur * ur + ui * ui
        rr, ri = rb[:, 0], rb[:, 1]
        ub[:, 0] = rr * (V + g * (rho + 2 * ur * ur) - lam) + ri * (2 * g * ur * ui)
        ub[:, 1] = ri * (V + g * (rho + 2 * ui * ui) - lam) + rr * (2 * g * ur * ui)
    # normalisation term  w_norm*(I-1)^2,  I = dx*sum u^2 (global)
    ub = ub + dt.type(pb.w_norm) * dt.type(4.0) * (I - 1) * dt.type(pb.dx) * u
    for j in range(n_orth):
        Oj = dt.type(tot[f'orth{j}'] * pb.dx)
        ub = ub + dt.type(pb.w_orth) * 2 * Oj * dt.type(pb.dx) * orth[j].astype(dt)[:, None]
    Ub = np.zeros_like(U)
    Ub[0] = ub
    c = dt.type(pb.kinetic_coeff)
    for k in range(d):
        Ub[1 + d + k] = -c * rb
    if pb.w_riesz != 0.0:
        ak, ap, ai, rz_norm = riesz_coefs(pb)
        dnm = dt.type(tot['den']) if rz_norm else dt.type(1.0)
        cI = 0.5 * (pb.p + 1) if pb.riesz_kind == RIESZ_VARIATIONAL else 1.0
        Eq = dt.type((tot['rz_k'] + tot['rz_p'] + cI * fI * tot['rz_i'] - pb.omega_rot * tot.get('rz_l', 0.0)) / tot['den']) if rz_norm else dt.type(0.0)
        wz = dt.type(pb.w_riesz)
        if not pb.complex_psi:
            dint = dt.type(ai * fI * (pb.p + 1)) * np.sign(u + (u == 0)) * _ipow(np.abs(u), pb.p)
        else:
            dint = dt.type(ai * fI * (pb.p + 1)) * (u * u).sum(axis=1, keepdims=True) * u          # d rho^2 / d psi_o = 4 rho psi_o (p = 3)
        Ub[0] = Ub[0] + wz * ((2 * dt.type(ap) * V[:, None] * u + dint) - 2 * Eq * u) / dnm
        for k in range(d):
            Ub[1 + k] = Ub[1 + k] + wz * dt.type(2 * ak) * U[1 + k] / dnm
        if pb.complex_psi and pb.omega_rot != 0.0:               # -Omega <L_z> / den: seeds on psi and on its first derivatives
            Om = dt.type(pb.omega_rot)
            xx, yy = x[:, 0].astype(dt), x[:, 1].astype(dt)
            Dr = xx * U[2, :, 0] - yy * U[1, :, 0]
            Di = xx * U[2, :, 1] - yy * U[1, :, 1]
            Ub[0, :, 0] += wz * (-Om * Di) / dnm
            Ub[0, :, 1] += wz * (Om * Dr) / dnm
            Ub[1, :, 0] += wz * (-Om * yy * u[:, 1]) / dnm       # d<L_z>/d(d_x psi_r) = y psi_i
            Ub[2, :, 0] += wz * (Om * xx * u[:, 1]) / dnm        # d<L_z>/d(d_y psi_r) = -x psi_i
            Ub[1, :, 1] += wz * (Om * yy * u[:, 0]) / dnm        # d<L_z>/d(d_x psi_i) = -y psi_r
            Ub[2, :, 1] += wz * (-Om * xx * u[:, 0]) / dnm       # d<L_z>/d(d_y psi_i) = x psi_r
    if pb.lambda_kind == LAMBDA_ENERGY or pb.w_reg_f != 0.0:
        # lambda is not the Rayleigh quotient of the residual's operator: d loss / d lambda = -2 w_pde / N sum r u (+ d L_lambda / d lambda)
        # reaches u and grad u through d lambda / d u (the branch SURVEY quirk Q10 says must be kept for the 2D classes)
        den = tot['den']
        if pb.lambda_kind == LAMBDA_ENERGY:
            lf = float(lam)
            lam_bar = -2.0 * pb.w_pde / N * (tot['num'] - lf * den)
            if pb.w_reg_lam != 0.0:
                lam_bar += -2.0 * pb.w_reg_lam * lf / (lf * lf + pb.reg_lam_eps) ** 2
            lb = dt.type(lam_bar / den)
            sgn = np.sign(u + (u == 0))
            Ub[0] = Ub[0] + lb * (2 * V[:, None] * u + g * dt.type(pb.p + 1) * sgn * _ipow(np.abs(u), pb.p) - 2 * lam * u)
            for k in range(d):
                Ub[1 + k] = Ub[1 + k] + lb * dt.type(2 * pb.kinetic_coeff) * U[1 + k]
        if pb.w_reg_f != 0.0:
            q = den / N + pb.reg_f_eps
            Ub[0] = Ub[0] + dt.type(-2.0 * pb.w_reg_f / (q * q * N)) * u
    if pb.complex_psi and pb.omega_rot != 0.0:
        Om = dt.type(pb.omega_rot)
        xx, yy = x[:, 0], x[:, 1]
        # Hr += -Om*(x psi_i,y - y psi_i,x) ; Hi += Om*(x psi_r,y - y psi_r,x)
        Ub[2, :, 1] += -Om * xx * rb[:, 0]
        Ub[1, :, 1] += Om * yy * rb[:, 0]
        Ub[2, :, 0] += Om * xx * rb[:, 1]
        Ub[1, :, 0] += -Om * yy * rb[:, 1]
    out_bar = dt.type(pb.perturb_scale) * Ub
    if pb.envelope == ENV_SIN:                                   # adjoint of psi = o f
        f, f1, f2 = (a[:, None] for a in envelope(pb, x[:, 0]))
        u0, u1, u2 = out_bar[0], out_bar[1], out_bar[2]
        out_bar = np.stack([f * u0 + f1 * u1 + f2 * u2, f * u1 + 2 * f1 * u2, f * u2])
    grad = mlp_backward(params, cache, out_bar, skip=skip).astype(acc)
    if x_bc is not None and pb.w_bc != 0.0:
        ebar = (dt.type(pb.w_bc * 2.0 / eb.size) * eb) * dt.type(pb.bc_nn_scale) * fenv_b
        gbc = mlp_backward(params, cb, ebar[None], value_only=True, skip=skip).astype(acc)
        res['grad_bc'] = gbc                                     # identical on every shard
    else:
        res['grad_bc'] = np.zeros_like(grad)
    if sym is not None:
        diff, c1, c2 = sym
        sb = dt.type(pb.w_sym * 2.0 / N) * diff
        grad = grad + mlp_backward(params, c1, sb[None], value_only=True, skip=skip)
        grad = grad + mlp_backward(params, c2, (-dt.type(pb.sym_sign) * sb)[None], value_only=True, skip=skip)
    res['grad_local'] = grad                                     # to be summed over shards
    res['psi'] = u
    res['residual'] = r
    return res


def assemble(pb: Problem, res: dict, sum_r2_total: Optional[float] = None, n_global: Optional[int] = None):
    """total loss = pde + w_bc*bc + w_norm*norm + w_sym*sym + w_orth*orth (:347,355; notebook c10:L97)."""
    N = n_global if n_global else (pb.n_global if pb.n_global > 0 else res['psi'].shape[0])
    sr2 = res['sum_r2'] if sum_r2_total is None else sum_r2_total
    pde = sr2 / N
    total = (pb.w_pde * pde + pb.w_bc * res['L_bc'] + pb.w_norm * res['L_norm']
             + pb.w_sym * res['L_sym'] + pb.w_orth * res['L_orth'] + pb.w_riesz * res.get('L_riesz', 0.0) + res.get('L_reg', 0.0))
    return dict(loss=total, pde=pde, bc=res['L_bc'], norm=res['L_norm'], sym=res['L_sym'],
                orth=res['L_orth'], mu=res['lam'], riesz=res.get('L_riesz', 0.0), reg=res.get('L_reg', 0.0))


def full_loss_and_grad(pb: Problem, flat, x, x_bc=None, bc_target=None, V_pre=None, orth=None, base_pre=None):
    res = loss_and_grad(pb, flat, x, x_bc, bc_target, V_pre, orth, base_pre=base_pre)
    sc = assemble(pb, res)
    grad = res['grad_local'] + res['grad_bc']
    return sc, grad, res


def sharded_loss_and_grad(pb: Problem, flat, x, x_bc=None, chunk: int = 65536, threads: int = 4, V_pre=None):
    """full_loss_and_grad of a batch too large to hold its jets at once (bench.py / tests: the oracle on the batch AS TIMED, 10^6
    points): the two-phase data-parallel protocol of loss_and_grad over contiguous chunks -- phase 1 adds the global sums (mu, norm
    integral), phase 2 the chunk gradients with those totals.  Same arithmetic as one call on the whole batch up to the order of
    the fp64 sums.  Chunks run on a small thread pool (numpy releases the GIL).  Problems without orthogonality targets (what the
    large workloads use); a precomputed potential is sliced with the points.  -> (scalars, gradient)"""
    from concurrent.futures import ThreadPoolExecutor
    import dataclasses
    N = x.shape[0]
    pbn = dataclasses.replace(pb, n_global=N)
    bounds = [(a, min(N, a + chunk)) for a in range(0, N, chunk)]
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        vp = (lambda a, b: None) if V_pre is None else (lambda a, b: V_pre[a:b])
        parts = list(ex.map(lambda ab: loss_and_grad(pbn, flat, x[ab[0]:ab[1]], V_pre=vp(*ab), phase=1), bounds))
        tot = {k: float(sum(p[k] for p in parts)) for k in parts[0]}
        # the boundary term is replicated: only the first chunk forms it
        res = list(ex.map(lambda ib: loss_and_grad(pbn, flat, x[ib[1][0]:ib[1][1]], x_bc if ib[0] == 0 else None, V_pre=vp(*ib[1]), shard_sums=tot),
                          enumerate(bounds)))
    r0 = res[0]
    sr2 = float(sum(r['sum_r2'] for r in res))
    sc = assemble(pbn, r0, sum_r2_total=sr2, n_global=N)
    grad = r0['grad_bc'].copy()
    for r in res:
        grad += r['grad_local']
    return sc, grad


# ----------------------------------------------------------------------------------------
# optimiser: clip_grad_norm_ + Adam + LR schedulers
# ----------------------------------------------------------------------------------------
@dataclass
class OptState:
    lr0: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    clip_norm: float = 1.0
    sched: int = SCHED_CONST
    # cosine warm restarts (harmonic_pinn_simulation.py:312-314)
    T_0: float = 200.0
    T_mult: float = 2.0
    eta_min: float = 1e-6
    # plateau (notebook c10:L76-78)
    factor: float = 0.5
    patience: int = 100
    min_lr: float = 1e-5
    threshold: float = 1e-4
    # state
    step: int = 0
    lr: float = field(default=None)
    best: float = float('inf')
    num_bad: int = 0
    m: Optional[np.ndarray] = None
    v: Optional[np.ndarray] = None

    def __post_init__(self):
        if self.lr is None:
            self.lr = self.lr0


def cosine_lr_from_loss(loss: float, lr0: float, T_0: float, T_mult: float, eta_min: float) -> float:
    """CosineAnnealingWarmRestarts.step(epoch=loss) -- quirk Q4 (harmonic_pinn_simulation.py:361):
    the LOSS VALUE is passed as the (fractional) epoch."""
    epoch = float(loss)
    if epoch >= T_0:
        if T_mult == 1:
            T_cur = epoch % T_0
            T_i = T_0
        else:
            n = int(math.log((epoch / T_0 * (T_mult - 1) + 1), T_mult))
            T_cur = epoch - T_0 * (T_mult ** n - 1) / (T_mult - 1)
            T_i = T_0 * T_mult ** n
    else:
        T_i = T_0
        T_cur = epoch
    return eta_min + (lr0 - eta_min) * (1 + math.cos(math.pi * T_cur / T_i)) / 2


def optimizer_step(st: OptState, flat: np.ndarray, grad: np.ndarray, loss: float, dtype=np.float32):
    """clip_grad_norm_(1.0) (:359), Adam defaults (:309,:360), scheduler.step(loss) (:361)."""
    dt = np.dtype(dtype).type
    g = grad.astype(dtype)
    gn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
    if st.clip_norm > 0:
        coef = min(1.0, st.clip_norm / (gn + 1e-6))
        g = g * dt(coef)
    if st.m is None:
        st.m = np.zeros_like(g)
        st.v = np.zeros_like(g)
    st.step += 1
    b1, b2 = st.beta1, st.beta2
    st.m = st.m + (g - st.m) * dt(1 - b1)                      # exp_avg.lerp_(grad, 1-beta1)
    st.v = st.v * dt(b2) + dt(1 - b2) * g * g
    bc1 = 1 - b1 ** st.step
    bc2 = 1 - b2 ** st.step
    step_size = st.lr / bc1
    denom = np.sqrt(st.v) / dt(math.sqrt(bc2)) + dt(st.eps)
    new = (flat.astype(dtype) - dt(step_size) * (st.m / denom)).astype(dtype)
    lr_used = st.lr
    # scheduler
    if st.sched == SCHED_COSINE_LOSS:
        st.lr = cosine_lr_from_loss(loss, st.lr0, st.T_0, st.T_mult, st.eta_min)
    elif st.sched == SCHED_PLATEAU:                            # ReduceLROnPlateau(mode='min', rel threshold)
        if loss < st.best * (1 - st.threshold):
            st.best = loss
            st.num_bad = 0
        else:
            st.num_bad += 1
        if st.num_bad > st.patience:
            new_lr = max(st.lr * st.factor, st.min_lr)
            if st.lr - new_lr > 1e-8:
                st.lr = new_lr
            st.num_bad = 0
    return new, gn, lr_used


def train_steps(pb: Problem, st: OptState, flat: np.ndarray, x, n_steps: int, x_bc=None, bc_target=None,
                V_pre=None, orth=None, dtype=np.float32, base_pre=None):
    """n_steps epochs of the reference loop body; returns params and per-step scalar trace."""
    x = x.astype(dtype)
    flat = flat.astype(dtype)
    trace = []
    for _ in range(n_steps):
        sc, grad, _ = full_loss_and_grad(pb, flat, x, x_bc, bc_target, V_pre, orth, base_pre)
        flat, gn, lr_used = optimizer_step(st, flat, grad, sc['loss'], dtype)
        sc.update(grad_norm=gn, lr=lr_used)
        trace.append(sc)
    return flat, trace


def eval_density(pb: Problem, flat: np.ndarray, x_test: np.ndarray, dx: float, abs_mode0: bool = False):
    """plot_wavefunction (:463-474) / notebook c12:L30-42: forward on the test grid, + base, renormalise."""
    dt = x_test.dtype
    params = unflatten(flat.astype(dt), pb.layers, pb.net_kind)
    _, skip, plain = expand_layers(pb.layers, pb.net_kind)
    o, _ = mlp_forward(params, x_test, pb.activation, value_only=True, skip=skip, plain_tanh=plain)
    u = dt.type(pb.perturb_scale) * o[0]
    if pb.envelope == ENV_SIN:
        u = u * envelope(pb, x_test[:, 0])[0][:, None]
    if pb.base_mode >= 0:
        u = u + base_functions(pb, x_test[:, 0])[0][:, None]
    nrm = np.sqrt((u * u).sum() * dt.type(dx))
    u = u / nrm
    if abs_mode0:
        u = np.abs(u)
    dens = (u * u).sum(axis=1)
    return u, dens
