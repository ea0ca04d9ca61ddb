"""The reference's Python surface for the hot path, re-hosted on the HIP engine.

Same class / function names, argument meaning and return shapes as
  refine flavour   : Gross-Pitaevskii/src/final/refine/harmonic_pinn_simulation.py  (class :52-217, train_gpe_model :220-430,
                     advanced_initialization :636-647, plot_wavefunction's normalisation :459-474)
  notebook flavour : Gross_Pitaevskii_1D_power_Test.ipynb  (class c6, train_gpe_model c10, advanced_initialization c18,
                     density c12:L30-42)
so that a caller of the reference can switch imports:

    from gpe_pinn.surface import refine          # refine.GrossPitaevskiiPINN, refine.train_gpe_model, ...
    from gpe_pinn.surface import notebook        # notebook.GrossPitaevskiiPINN, notebook.train_gpe_model, ...

Every loss / derivative / optimiser number comes from libgpe_hip.so (C ABI, include/gpe_hip.h).  torch is used for
device storage, for the SAME host RNG the reference uses in its weight init (so seeded runs start from identical
weights) and for torch.distributed.  There is no autograd and no CPU compute path here.
"""
from __future__ import annotations

import math
import types
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _capi as capi
from .engine import Engine, GPEConfig

_POT = {"harmonic": capi.POT_HARMONIC, "gaussian": capi.POT_GAUSSIAN, "periodic": capi.POT_PERIODIC, "box": capi.POT_NONE,
        "gravity_well": capi.POT_PRECOMPUTED}


def _device():
    return torch.device("cuda", torch.cuda.current_device())


class _Flavor:
    def __init__(self, name, activation, kinetic, pot_scale, base_deriv, w_sym, init_kind):
        self.name, self.activation, self.kinetic, self.pot_scale = name, activation, kinetic, pot_scale
        self.base_deriv, self.w_sym, self.init_kind = base_deriv, w_sym, init_kind


_REFINE = _Flavor("refine", capi.ACT_TANH_PLUS1, 1.0, 1.0, 0, 0.0, "xavier_normal")
_NOTEBOOK = _Flavor("notebook", capi.ACT_TANH, 0.5, 0.5, 1, 5.0, "xavier_uniform")


def _layer_shapes(layers):
    return [(layers[i + 1], layers[i]) for i in range(len(layers) - 1)]


def _mlp_spec(layers):
    """[(state_dict key prefix, out, in)] of nn.Sequential(Linear, act, Linear, ...): keys network.{2k}"""
    return [(f"network.{2 * k}", fo, fi) for k, (fo, fi) in enumerate(_layer_shapes(layers))]


def _residual_spec(layers):
    """refine/box_to_gaussian_pinn_simulation.py:112-130: network.0 = Linear(d,H), network.1 = ShiftedTanh, network.{2+b} =
    ResidualBlock b (lin1, lin2), network.{2+nb} = Linear(H,out)"""
    d, H, out, nb = layers[0], layers[1], layers[-1], len(layers) - 3
    spec = [("network.0", H, d)]
    for b in range(nb):
        spec += [(f"network.{2 + b}.lin1", H, H), (f"network.{2 + b}.lin2", H, H)]
    return spec + [(f"network.{2 + nb}", out, H)]


def _state_dict_from_flat(flat: np.ndarray, layers, spec=None) -> "OrderedDict[str, torch.Tensor]":
    """keys network.{2k}.weight / .bias as nn.Sequential(Linear, act, Linear, ...) gives them."""
    sd, o = OrderedDict(), 0
    for key, fo, fi in (spec or _mlp_spec(layers)):
        sd[f"{key}.weight"] = torch.from_numpy(flat[o:o + fo * fi].reshape(fo, fi).copy()); o += fo * fi
        sd[f"{key}.bias"] = torch.from_numpy(flat[o:o + fo].copy()); o += fo
    return sd


def _flat_from_state_dict(sd, layers, spec=None) -> np.ndarray:
    parts = []
    for key, fo, fi in (spec or _mlp_spec(layers)):
        k = key
        W = sd[f"{key}.weight"]
        b = sd[f"{key}.bias"]
        W = W.detach().cpu().numpy() if isinstance(W, torch.Tensor) else np.asarray(W)
        b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
        if W.shape != (fo, fi) or b.shape != (fo,):
            raise RuntimeError(f"size mismatch for {k}: got {W.shape}/{b.shape}, expected {(fo, fi)}/{(fo,)}")
        parts += [W.astype(np.float32).ravel(), b.astype(np.float32).ravel()]
    return np.concatenate(parts)


def _default_init(layers, spec=None) -> np.ndarray:
    """nn.Linear's default init (kaiming_uniform(a=sqrt 5) + uniform bias), drawn from torch's global CPU RNG in the same
    order the reference's constructor draws it (refine/...:90)."""
    parts = []
    for fo, fi in ([(a, b) for _, a, b in spec] if spec else _layer_shapes(layers)):
        lin = torch.nn.Linear(fi, fo)
        parts += [lin.weight.detach().numpy().ravel(), lin.bias.detach().numpy().ravel()]
    return np.concatenate(parts).astype(np.float32)


def _advanced_init(layers, mode, kind, spec=None) -> np.ndarray:
    """advanced_initialization: refine/...:636-647 (Xavier-normal, gain 1/(1+0.2 mode), bias 0.01 | 0.001 if mode>3);
    nb c18 (Xavier-uniform, gain 1/(1+0.1 mode), bias 0.01).  Uses torch's CPU RNG exactly as model.apply() does."""
    parts = []
    for fo, fi in ([(a, b) for _, a, b in spec] if spec else _layer_shapes(layers)):
        W = torch.empty(fo, fi)
        if kind == "xavier_normal":
            torch.nn.init.xavier_normal_(W, gain=1.0 / (1.0 + 0.2 * mode))
            bias = 0.001 if mode > 3 else 0.01
        elif kind == "xavier_normal_vbeta":                  # refine/vary_potential_parameter_harmonic.py:788-798 (and its two siblings)
            torch.nn.init.xavier_normal_(W, gain=(0.1 if mode >= 3 else 1.0) / (1.0 + 0.2 * mode))
            bias = 0.0001 if mode >= 3 else 0.01
        else:
            torch.nn.init.xavier_uniform_(W, gain=1.0 / (1.0 + 0.1 * mode))
            bias = 0.01
        parts += [W.numpy().ravel(), np.full(fo, bias, np.float32)]
    return np.concatenate(parts).astype(np.float32)


def seeded_reference_init(layers, mode, kind) -> np.ndarray:
    """The weights a seeded reference run starts from: GrossPitaevskiiPINN(layers) draws nn.Linear's default init from the
    global RNG, then model.apply(advanced_initialization) overwrites it (refine/...:295-304 ; nb c10:L63-70)."""
    _default_init(layers)
    return _advanced_init(layers, mode, kind)


class _PINNBase:
    """Parameter container + per-call loss surface.  The training loop does not go through these methods (it drives
    Engine.step / Engine.run); they exist so that code written against the reference class keeps working."""
    _flavor: _Flavor = None

    def __init__(self, layers, hbar=1.0, m=1.0, mode=0, gamma=1.0):
        self.layers = list(layers)
        self.hbar, self.m, self.mode, self.gamma = hbar, m, mode, gamma
        self._flat = _default_init(self.layers, self._spec())     # nn.Linear default init, as the reference constructor does
        self._engine: Optional[Engine] = None
        self._engine_key = None
        self.perturb_scale = 1.0

    # -- nn.Module-like conveniences ---------------------------------------------------------------
    def to(self, device=None):
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def cpu(self):
        return self

    def apply(self, fn):
        """model.apply(lambda m: advanced_initialization(m, mode)): the initialiser is applied to the whole flat vector."""
        fn(self)
        return self

    def parameters(self):
        return [torch.from_numpy(self._flat)]

    def _spec(self):
        return None                                        # plain MLP: network.{2k}

    def state_dict(self):
        self._pull()
        return _state_dict_from_flat(self._flat, self.layers, self._spec())

    def load_state_dict(self, sd):
        self._flat = _flat_from_state_dict(sd, self.layers, self._spec())
        if self._engine is not None:
            self._engine.set_params(self._flat)

    def _pull(self):
        if self._engine is not None:
            self._flat = self._engine.get_params()

    # -- engine management ---------------------------------------------------------------------------------
    def _config(self, **over) -> GPEConfig:
        f = self._flavor
        kw = dict(layers=self.layers, activation=f.activation, kinetic_coeff=f.kinetic, pot_scale=f.pot_scale,
                  base_mode=(self.mode if getattr(self, "use_perturbation", True) else -1), base_deriv=f.base_deriv,
                  gamma=float(self.gamma), perturb_scale=float(self.perturb_scale), w_sym=f.w_sym,
                  sym_sign=(-1.0 if self.mode % 2 == 1 else 1.0))
        kw.update(self._extra_config())
        kw.update(over)
        return GPEConfig(**kw)

    def _extra_config(self) -> dict:
        return {}

    def _bind_training_data(self, eng, X_np, X_dev, bpts):
        """X_tensor / boundary_points of the training loop (refine/...:260-264)."""
        eng.bind_points(X_dev)
        eng.bind_boundary(bpts)

    def _get_engine(self, **over) -> Engine:
        key = tuple(sorted(over.items()))
        if self._engine is None or key != self._engine_key:
            if self._engine is not None:
                self._pull()
                self._engine.close()
            self._engine = Engine(self._config(**over))
            self._engine.set_params(self._flat)
            self._engine_key = key
        return self._engine

    # -- reference methods ---------------------------------------------------------------------------------------
    def forward(self, inputs):
        return self._get_engine().forward(inputs)

    __call__ = forward

    def weighted_hermite(self, x, n):
        """phi_n(x) (refine/...:95-119): evaluated by the engine's eval path with a zero perturbation."""
        eng = self._get_engine()
        x2 = x.reshape(-1, 1)
        old = eng.cfg.perturb_scale
        cfgm = eng.cfg.base_mode
        if cfgm != n:
            tmp = Engine(self._config(base_mode=int(n), perturb_scale=0.0))
            tmp.set_params(self._flat)
            sc, psi, _ = self._psi_only(tmp, x2)
            tmp.close()
            return psi.reshape(x.shape)
        eng.set_perturb_scale(0.0)
        sc, psi, _ = self._psi_only(eng, x2)
        eng.set_perturb_scale(old)
        return psi.reshape(x.shape)

    @staticmethod
    def _psi_only(eng, x2):
        eng.bind_points(x2)
        return eng.residual()

    @staticmethod
    def weighted_hermite_np(x, n):
        """phi_n on the host in fp64 (target of the pre-training fit)."""
        x = np.asarray(x, dtype=np.float64)
        Hm1, Hc = np.zeros_like(x), np.ones_like(x)
        for k in range(n):
            Hm1, Hc = Hc, 2 * x * Hc - 2 * k * Hm1
        norm = ((2.0 ** n) * float(math.factorial(n)) * math.sqrt(math.pi)) ** (-0.5)
        return (norm * Hc * np.exp(-0.5 * x * x)).astype(np.float32)

    def get_complete_solution(self, x, perturbation, mode=None):
        mode = self.mode if mode is None else mode
        return self.weighted_hermite(x, mode) + perturbation

    def compute_potential(self, x, potential_type="harmonic", **kwargs):
        if potential_type not in _POT:
            raise ValueError(f"Unknown potential type: {potential_type}")
        f = self._flavor
        if potential_type == "harmonic":
            omega = kwargs.get("omega", 1.0)
            if "beta" in kwargs or "center" in kwargs:       # refine/vary_potential_parameter_harmonic.py:231-240
                return kwargs.get("beta", 1.0) * 0.5 * omega ** 2 * (x - kwargs.get("center", 0.0)) ** 2
            return f.pot_scale * (omega * x) ** 2 if f is _NOTEBOOK else x ** 2
        if potential_type == "gaussian":
            return torch.exp(-(x - kwargs.get("a", 0.0)) ** 2)
        V0, k = kwargs.get("V0", 1.0), kwargs.get("k", 2 * np.pi / 5.0)
        return V0 * torch.cos(k * x) ** 2

    def _scale_of(self, inputs, predictions):
        """The reference passes `predictions` = s * forward(inputs) (s = q/normal_const or 1); recover s."""
        nn_out = self.forward(inputs)
        den = float((nn_out * nn_out).sum())
        return float((predictions.detach() * nn_out).sum()) / den if den > 0 else 1.0

    def _pde_eval(self, inputs, predictions, gamma, p, potential_type, precomputed_potential, **over):
        if precomputed_potential is None and potential_type not in _POT:
            raise ValueError(f"Unknown potential type: {potential_type}")
        pot = capi.POT_PRECOMPUTED if precomputed_potential is not None else _POT[potential_type]
        s = self._scale_of(inputs, predictions)
        eng = self._get_engine(potential=pot, p=int(p), w_bc=0.0, **over)
        eng.set_gamma(float(gamma))
        eng.set_perturb_scale(s)
        eng.bind_points(inputs, precomputed_potential)
        self._before_residual(eng, inputs)
        return eng.residual()

    def _before_residual(self, eng, inputs):
        pass

    def boundary_loss(self, boundary_points, boundary_values):
        eng = self._get_engine()
        if not hasattr(eng, "n_local") or eng.cfg.potential == capi.POT_PRECOMPUTED or eng.cfg.base_kind == capi.BASE_PRECOMPUTED:
            # the residual launch needs SOME collocation batch: the boundary points themselves (flavours whose potential / base function
            # are handed over as arrays get arrays of that length; the boundary term uses neither)
            V = torch.zeros(boundary_points.shape[0], dtype=torch.float32, device=boundary_points.device) \
                if eng.cfg.potential == capi.POT_PRECOMPUTED else None
            eng.bind_points(boundary_points, V)
            self._before_residual(eng, boundary_points)
        eng.bind_boundary(boundary_points, self._boundary_target(boundary_points, boundary_values))
        sc, _, _ = eng.residual(want_fields=False)
        return torch.tensor(sc["bc"], dtype=torch.float32, device=_device())

    def _boundary_target(self, boundary_points, boundary_values):
        return boundary_values

    def normalization_loss(self, u, dx):
        integral = torch.sum(u ** 2) * dx
        return (integral - 1.0) ** 2

    def close(self):
        if self._engine is not None:
            self._pull()
            self._engine.close()
            self._engine = None


class _RefinePINN(_PINNBase):
    """refine/harmonic_pinn_simulation.py:52-217."""
    _flavor = _REFINE

    def __init__(self, layers, hbar=1.0, m=1.0, mode=0, gamma=1.0, use_perturbation=True):
        super().__init__(layers, hbar, m, mode, gamma)
        self.use_perturbation = use_perturbation

    def pde_loss(self, inputs, predictions, gamma, p, potential_type="harmonic", precomputed_potential=None):
        sc, _, _ = self._pde_eval(inputs, predictions, gamma, p, potential_type, precomputed_potential)
        dev = _device()
        return (torch.tensor(sc["pde"], dtype=torch.float32, device=dev),
                torch.tensor(sc["mu"], dtype=torch.float32, device=dev))


class _BoxPINN(_RefinePINN):
    """refine/box_pinn_simulation.py:52-265: particle in a box [0, L]; forward = network(x) * sin(pi x) (hard boundary
    factor, :119-130), base sqrt(2/L) sin((n+1) pi x / L) (:99-117), V = 0."""

    def __init__(self, layers, hbar=1.0, m=1.0, mode=0, gamma=1.0, L=1.0, use_residual=True, use_perturbation=True):
        super().__init__(layers, hbar, m, mode, gamma, use_perturbation)
        self.L = L
        self.use_residual = use_residual

    def _extra_config(self):
        return dict(base_kind=capi.BASE_BOX, box_L=float(self.L), envelope=capi.ENV_SIN, env_L=1.0, potential=capi.POT_NONE)

    def box_eigenfunction(self, x, n):
        return math.sqrt(2.0 / self.L) * torch.sin((n + 1) * math.pi * x / self.L)

    def weighted_hermite(self, x, n):            # the base of this flavour
        return self.box_eigenfunction(x, n)

    def weighted_hermite_np(self, x, n):
        return (math.sqrt(2.0 / self.L) * np.sin((n + 1) * math.pi * np.asarray(x, np.float64) / self.L)).astype(np.float32)

    def compute_potential(self, x, potential_type="box", **kwargs):
        if potential_type != "box":
            raise ValueError(f"Unknown potential type: {potential_type}")
        return torch.zeros_like(x)

    def pde_loss(self, inputs, predictions, gamma, p, potential_type="box", precomputed_potential=None):
        return super().pde_loss(inputs, predictions, gamma, p, potential_type, precomputed_potential)


class _BoxToGaussianPINN(_RefinePINN):
    """refine/box_to_gaussian_pinn_simulation.py:66-240: box [0, L] with the Gaussian bump V = exp(-(x - 0.5)^2) inside; sine base
    sqrt(2/L) sin((n+1) pi x / L) (:132-150); forward = network(x) (no boundary factor, :152-156); the network is
    Linear + ShiftedTanh, len(layers)-3 ResidualBlocks tanh(lin2(tanh(lin1 x)) + x), Linear (:100-130) when use_residual (default),
    else the plain ShiftedTanh MLP."""

    def __init__(self, layers, hbar=1.0, m=1.0, mode=0, gamma=1.0, L=1.0, use_residual=True):
        self.use_residual = bool(use_residual)
        self.L = L
        # the reference constructor builds its network TWICE (refine/box_to_gaussian_pinn_simulation.py:90 and :98): two default-init
        # draws from the global RNG before advanced_initialization -- consume the first one here so that seeded runs start alike
        _default_init(list(layers), _residual_spec(list(layers)) if self.use_residual else None)
        super().__init__(layers, hbar, m, mode, gamma, True)

    def _spec(self):
        return _residual_spec(self.layers) if self.use_residual else None

    def _extra_config(self):
        return dict(base_kind=capi.BASE_BOX, box_L=float(self.L), potential=capi.POT_GAUSSIAN, pot_a=0.5,
                    net_kind=capi.NET_RESIDUAL if self.use_residual else capi.NET_MLP)

    def box_eigenfunction(self, x, n):
        return math.sqrt(2.0 / self.L) * torch.sin((n + 1) * math.pi * x / self.L)

    def weighted_hermite(self, x, n):            # the base of this flavour
        return self.box_eigenfunction(x, n)

    def weighted_hermite_np(self, x, n):
        return (math.sqrt(2.0 / self.L) * np.sin((n + 1) * math.pi * np.asarray(x, np.float64) / self.L)).astype(np.float32)

    def compute_potential(self, x, potential_type="gaussian", **kwargs):
        if potential_type != "gaussian":
            raise ValueError(f"Unknown potential type: {potential_type}")
        return torch.exp(-(x - 0.5) ** 2)

    def pde_loss(self, inputs, predictions, gamma, p, potential_type="gaussian", precomputed_potential=None):
        return super().pde_loss(inputs, predictions, gamma, p, potential_type, precomputed_potential)


class _GravityWellPINN(_RefinePINN):
    """refine/gravity_well_pinn_simulation.py:52-260: V = x on [0, ub]; base Psi_n = Ai(x + alpha_n) normalised on the grid,
    Psi' from scipy's Ai', Psi'' by np.gradient (as the reference does, :141-173).  Like the reference, the base is computed on
    the host with scipy and handed to the device as arrays (GPE_BASE_PRECOMPUTED); V = x goes in as a precomputed potential."""

    def _extra_config(self):
        return dict(base_kind=capi.BASE_PRECOMPUTED, potential=capi.POT_PRECOMPUTED)

    @staticmethod
    def _airy(x_np, n):
        from scipy.special import ai_zeros, airy
        alpha_n = ai_zeros(n + 1)[0][n]
        return airy(x_np + alpha_n)

    def airy_solution(self, x, n):
        """:97-118 (normalised with dx of the points handed in -- including the two boundary points, quirk kept)."""
        x_np = _as_np(x).astype(np.float64).reshape(-1)
        psi = self._airy(x_np, n)[0]
        dx = x_np[1] - x_np[0] if len(x_np) > 1 else 0.01
        psi = psi / np.sqrt(np.sum(psi ** 2) * dx)
        out = torch.tensor(psi.astype(np.float32), device=_device())
        return out.reshape(x.shape) if isinstance(x, torch.Tensor) else out

    def base_arrays(self, x_np, n):
        """(phi, phi', phi'') exactly as get_complete_solution_with_derivatives builds them (:141-160)."""
        x_np = np.asarray(x_np, np.float64).reshape(-1)
        ai, aip, _, _ = self._airy(x_np, n)
        dx = x_np[1] - x_np[0] if len(x_np) > 1 else 0.01
        nf = np.sqrt(np.sum(ai ** 2) * dx)
        return ((ai / nf).astype(np.float32), (aip / nf).astype(np.float32), np.gradient(aip / nf, dx).astype(np.float32))

    def weighted_hermite(self, x, n):
        return self.airy_solution(x, n)

    def weighted_hermite_np(self, x, n):
        return self.base_arrays(x, n)[0]

    def compute_potential(self, x, potential_type="gravity_well", **kwargs):
        if potential_type != "gravity_well":
            raise ValueError(f"Unknown potential type: {potential_type}")
        return x.clone()

    def _before_residual(self, eng, inputs):               # per-call surface: the Airy base on the points just bound
        eng.bind_base(*self.base_arrays(_as_np(inputs)[:, 0], self.mode))

    def _boundary_target(self, boundary_points, boundary_values):      # e = base(x_b) + NN(x_b) - values: the array base goes into the target
        return boundary_values - self.airy_solution(boundary_points, self.mode).reshape(boundary_values.shape)

    def pde_loss(self, inputs, predictions, gamma, p, potential_type="gravity_well", precomputed_potential=None):
        if precomputed_potential is None:
            precomputed_potential = self.compute_potential(inputs, potential_type).reshape(-1).contiguous()      # V = x, as an array
        return super().pde_loss(inputs, predictions, gamma, p, "gravity_well", precomputed_potential)

    def _bind_training_data(self, eng, X_np, X_dev, bpts):
        eng.bind_points(X_dev, V=X_dev[:, 0].contiguous())
        eng.bind_base(*self.base_arrays(X_np[:, 0], self.mode))
        base_b = self.airy_solution(bpts, self.mode).reshape(-1, 1)
        eng.bind_boundary(bpts, -base_b)               # e = base(x_b) + NN(x_b) - 0


# ---- beta-sweep flavours: refine/vary_potential_parameter_{harmonic,gravity_well,box_and_gaussian}.py --------------------------
# Same residual and epoch body as the refine family; what changes is the constructor (beta instead of gamma), that the potential
# term carries beta, the signature pde_loss(inputs, predictions, gamma, beta, p, ...) and the driver's loop: ONE interaction strength,
# continuation in beta.
class _VaryBetaMixin:
    def _init_beta(self, beta, use_residual):
        self.beta = beta
        self.use_residual = use_residual
        self.gamma = 0.0                                   # set by the driver (train_gpe_model(gamma, beta_values, ...))

    def _beta_potential(self, X_np, potential_type):       # beta * V on the training grid, or None when the engine forms it
        return None

    def _pde_over(self, beta):                             # engine overrides that carry beta when the engine forms the potential
        return {}


class _VaryHarmonicPINN(_VaryBetaMixin, _RefinePINN):
    """refine/vary_potential_parameter_harmonic.py:52-342: box [0, L] with the trap V = beta/2 omega^2 (x - center)^2 inside
    (omega = 10, center = 2.5: :236-238); sine base sqrt(2/L) sin((n+1) pi x / L) (:99-117); forward = network(x)."""
    OMEGA, CENTER = 10.0, 2.5

    def __init__(self, layers, hbar=1.0, m=1.0, mode=0, beta=1.0, L=5.0, use_residual=True, use_perturbation=True):
        _RefinePINN.__init__(self, layers, hbar, m, mode, 0.0, use_perturbation)
        self._init_beta(beta, use_residual)
        self.L = L

    def _extra_config(self):
        return dict(base_kind=capi.BASE_BOX, box_L=float(self.L), potential=capi.POT_HARMONIC, pot_scale=0.5 * float(self.beta),
                    omega=(self.OMEGA, 1.0, 1.0), pot_a=self.CENTER)

    def _pde_over(self, beta):
        return dict(pot_scale=0.5 * float(beta))

    def box_eigenfunction(self, x, n):
        return math.sqrt(2.0 / self.L) * torch.sin((n + 1) * math.pi * x / self.L)

    def weighted_hermite_np(self, x, n):                   # target of the pre-training fit of this script: the box eigenfunction (:807)
        return (math.sqrt(2.0 / self.L) * np.sin((n + 1) * math.pi * np.asarray(x, np.float64) / self.L)).astype(np.float32)

    def get_complete_solution(self, x, perturbation, mode=None):
        return self.box_eigenfunction(x, self.mode if mode is None else mode) + perturbation

    def compute_potential(self, x, beta, potential_type="harmonic", **kwargs):
        if potential_type != "harmonic":
            raise ValueError(f"Unknown potential type: {potential_type}")
        return beta * 0.5 * kwargs.get("omega", self.OMEGA) ** 2 * (x - kwargs.get("center", self.CENTER)) ** 2

    def pde_loss(self, inputs, predictions, gamma, beta, p, potential_type="harmonic", precomputed_potential=None):
        if precomputed_potential is None and potential_type != "harmonic":
            raise ValueError(f"Unknown potential type: {potential_type}")
        sc, _, _ = self._pde_eval(inputs, predictions, gamma, p, "harmonic", precomputed_potential, **self._pde_over(beta))
        dev = _device()
        return (torch.tensor(sc["pde"], dtype=torch.float32, device=dev), torch.tensor(sc["mu"], dtype=torch.float32, device=dev))


class _VaryGravityPINN(_VaryBetaMixin, _GravityWellPINN):
    """refine/vary_potential_parameter_gravity_well.py:52-257: the gravity well with the potential term beta * x * u (:221)."""

    def __init__(self, layers, hbar=1.0, m=1.0, mode=0, beta=1.0, use_residual=True, use_perturbation=True):
        _GravityWellPINN.__init__(self, layers, hbar, m, mode, 0.0, use_perturbation)
        self._init_beta(beta, use_residual)

    def _beta_potential(self, X_np, potential_type):
        return (float(self.beta) * np.asarray(X_np, np.float64)[:, 0]).astype(np.float32)

    def _bind_training_data(self, eng, X_np, X_dev, bpts):
        super()._bind_training_data(eng, X_np, X_dev, bpts)
        eng.bind_points(X_dev, V=torch.as_tensor(self._beta_potential(X_np, "gravity_well"), device=X_dev.device))

    def pde_loss(self, inputs, predictions, gamma, beta, p, potential_type="gravity_well", precomputed_potential=None):
        V = precomputed_potential if precomputed_potential is not None else self.compute_potential(inputs, potential_type)
        sc, _, _ = self._pde_eval(inputs, predictions, gamma, p, "gravity_well", (float(beta) * V).reshape(-1).contiguous())
        dev = _device()
        return (torch.tensor(sc["pde"], dtype=torch.float32, device=dev), torch.tensor(sc["mu"], dtype=torch.float32, device=dev))


class _VaryBoxGaussianPINN(_VaryBetaMixin, _BoxPINN):
    """refine/vary_potential_parameter_box_and_gaussian.py:52-225: box [0, L], forward = network(x) sin(pi x) (:119-130), sine base, the
    potential term beta * V * u with V = exp(-x^2 / 2) ("gaussian", :147) or 0 ("box")."""

    def __init__(self, layers, hbar=1.0, m=1.0, mode=0, beta=1.0, L=1.0, use_residual=True, use_perturbation=True):
        _BoxPINN.__init__(self, layers, hbar, m, mode, 0.0, L, use_residual, use_perturbation)
        self._init_beta(beta, use_residual)

    def _extra_config(self):
        return dict(base_kind=capi.BASE_BOX, box_L=float(self.L), envelope=capi.ENV_SIN, env_L=1.0, potential=capi.POT_PRECOMPUTED)

    def compute_potential(self, x, potential_type="gaussian", **kwargs):
        if potential_type == "box":
            return torch.zeros_like(x)
        if potential_type == "gaussian":
            return torch.exp((-x ** 2) / 2)
        raise ValueError(f"Unknown potential type: {potential_type}")

    def _beta_potential(self, X_np, potential_type):
        x = np.asarray(X_np, np.float64)[:, 0]
        if potential_type == "box":
            return np.zeros_like(x, dtype=np.float32)
        if potential_type != "gaussian":
            raise ValueError(f"Unknown potential type: {potential_type}")
        return (float(self.beta) * np.exp(-x * x / 2)).astype(np.float32)

    def _bind_training_data(self, eng, X_np, X_dev, bpts):
        eng.bind_points(X_dev, V=torch.as_tensor(self._beta_potential(X_np, self._pot_name), device=X_dev.device))
        eng.bind_boundary(bpts)

    def pde_loss(self, inputs, predictions, gamma, beta, p, potential_type="gaussian", precomputed_potential=None):
        V = precomputed_potential if precomputed_potential is not None else self.compute_potential(inputs, potential_type)
        sc, _, _ = self._pde_eval(inputs, predictions, gamma, p, "box", (float(beta) * V).reshape(-1).contiguous())
        dev = _device()
        return (torch.tensor(sc["pde"], dtype=torch.float32, device=dev), torch.tensor(sc["mu"], dtype=torch.float32, device=dev))


class _NotebookPINN(_PINNBase):
    """Gross_Pitaevskii_1D_power_Test.ipynb c6."""
    _flavor = _NOTEBOOK

    def __init__(self, layers, hbar=1.0, m=1.0, mode=0, gamma=1.0, power=1.0):
        super().__init__(layers, hbar, m, mode, gamma)
        self.power = power
        self.use_perturbation = True

    def pde_loss(self, p, inputs, predictions, gamma, potential_type="harmonic", precomputed_potential=None):
        sc, psi, res = self._pde_eval(inputs, predictions, gamma, p, potential_type, precomputed_potential)
        dev = _device()
        return (torch.tensor(sc["pde"], dtype=torch.float32, device=dev), res,
                torch.tensor(sc["mu"], dtype=torch.float32, device=dev), psi)

    def symmetry_loss(self, collocation_points, lb, ub):
        eng = self._get_engine()
        eng.bind_points(collocation_points)
        sc, _, _ = eng.residual(want_fields=False)
        return torch.tensor(sc["sym"], dtype=torch.float32, device=_device())


def _as_np(X):
    return X.detach().cpu().numpy() if isinstance(X, torch.Tensor) else np.asarray(X)


class _History:
    """Every epoch's record of one driver stage, kept on the returned model (`model.history[k]["loss" | "mu" | "lr" | ...]`): column
    arrays instead of one dict per epoch (the 201-stage experiment records 400 000 epochs).  Built from a list of records or from the
    engine's [n, fields] array."""

    def __init__(self, rows, keys=None):
        if isinstance(rows, np.ndarray):
            self._keys = list(keys)
            self._cols = {k: rows[:, i] for i, k in enumerate(self._keys)}
            self._n = rows.shape[0]
        else:
            self._keys = list(rows[0].keys()) if rows else []
            self._cols = {k: np.array([r[k] for r in rows], dtype=np.float64) for k in self._keys}
            self._n = len(rows)

    def __len__(self):
        return self._n

    def __bool__(self):
        return self._n > 0

    def __getitem__(self, k):
        if isinstance(k, str):
            return self._cols[k]
        if k < 0:
            k += self._n
        if not 0 <= k < self._n:
            raise IndexError(k)
        return {key: float(col[k]) for key, col in self._cols.items()}

    def __iter__(self):
        return (self[k] for k in range(self._n))


def _history(eng: Engine, first: int, last: int, chunk: int = 8192) -> "_History":
    parts = []
    s = first
    while s <= last:
        n = min(chunk, last - s + 1)
        parts.append(eng.read_history_array(s, n))
        s += n
    arr = np.concatenate(parts) if parts else np.zeros((0, len(Engine.HISTORY_FIELDS)))
    return _History(arr, Engine.HISTORY_FIELDS)


# ================================================================================================
# refine flavour driver
# ================================================================================================
def _refine_advanced_initialization(m, mode):
    m._flat = _advanced_init(m.layers, mode, "xavier_normal", m._spec())
    if m._engine is not None:
        m._engine.set_params(m._flat)


def _refine_pretrain(model, mode, X_train, epochs=5000, lr=1e-3, verbose=False):
    """pretrain_on_analytical_solution (refine/harmonic_pinn_simulation.py:650-701): fit the network output to phi_mode(x).
    Adam phase (epochs-500 iterations, plain Adam, no clipping) on the engine; then the reference's L-BFGS tail
    (torch.optim.LBFGS(lr*0.1, max_iter=20), 500 outer iterations) driven host-side on the engine's loss/gradient."""
    X = _as_np(X_train).astype(np.float32)
    dev = _device()
    X_dev = torch.as_tensor(X, device=dev)
    # the fit involves no physics: no base, no potential (only the hard boundary factor of forward(), if any, stays)
    eng = model._get_engine(lr=float(lr), sched=capi.SCHED_CONST, w_bc=0.0, potential=capi.POT_NONE, base_mode=-1)
    eng.reset_optimizer(float(lr))
    eng.bind_points(X_dev)
    target = model.weighted_hermite_np(X[:, 0], mode).reshape(-1, 1)
    eng.bind_target(torch.as_tensor(target, device=dev))
    loss = float("inf")
    n_adam = max(epochs - 500, 0)
    for epoch in range(n_adam):
        loss = eng.mse_step()["loss"]
        if verbose and epoch % 500 == 0:
            print(f"  Pre-training epoch {epoch}, loss: {loss:.2e}")
        if loss < 1e-12:
            break
    if loss >= 1e-12 and epochs > n_adam:
        theta = torch.tensor(eng.get_params(), dtype=torch.float32, requires_grad=True)
        opt = torch.optim.LBFGS([theta], lr=lr * 0.1, max_iter=20)

        def closure():
            opt.zero_grad()
            eng.set_params(theta.detach().numpy())
            l, g = eng.mse_loss_grad()
            theta.grad = torch.from_numpy(g.copy())
            return torch.tensor(l, dtype=torch.float32)

        for epoch in range(n_adam, epochs):
            loss = float(opt.step(closure))
            if loss < 1e-12:
                break
        eng.set_params(theta.detach().numpy())
    if verbose:
        print(f"  Pre-training finished, final loss: {loss:.2e}")
    eng.bind_target(None)
    model._pull()
    model.pretrain_loss = loss
    return model


def _refine_train(gamma_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const,
                  potential_type="harmonic", lr=1e-5, verbose=True, pretrain="reference", chunk=128, _cls=None, _descending=False,
                  _make_model=None, _init_fn=None, _pot_over=None, _label="γ", **model_kw):
    """train_gpe_model of refine/harmonic_pinn_simulation.py:220-430 (PL-PINN, gamma continuation).

    Differences from the reference, all deliberate:
      Q5  the reference "restores the best model" from a shallow state_dict copy, i.e. it keeps the LAST weights; so do we.
      Q6  normal_const is taken at the first epoch of the first gamma of each mode (the reference only defines it when that
          gamma is 0 and raises NameError otherwise).
      pretraining (:300-303): `pretrain="reference"` (default) runs pretrain_on_analytical_solution(epochs=2000, lr=1e-3) when
      the first gamma is 0, as the reference does; `pretrain=None` skips it (advanced_initialization instead); a callable
      (model, mode, X_train) -> model replaces it.
    Epoch bodies are enqueued `chunk` at a time with no host synchronisation; early stopping (:389-400) is evaluated on the
    device after every update, so the stop epoch is exact.
    """
    if _pot_over is None and potential_type not in _POT:
        raise ValueError(f"Unknown potential type: {potential_type}")
    # (the beta-sweep drivers below reuse this loop: `gamma_values` is then the list of beta values, _make_model builds the model of a
    # stage, _pot_over replaces the potential override of the engine configuration, _init_fn the cold-start initialiser)
    pot_over = {"potential": _POT[potential_type]} if _pot_over is None else dict(_pot_over)
    init_fn = _init_fn or _refine_advanced_initialization
    X = _as_np(X_train).astype(np.float64)
    dx = X[1, 0] - X[0, 0]
    dev = _device()
    X_dev = torch.as_tensor(X.astype(np.float32), device=dev)
    bpts = torch.tensor([[lb], [ub]], dtype=torch.float32, device=dev)
    models_by_mode, mu_table, training_history, constant_history, epochs_history = {}, {}, {}, {}, {}
    gamma_values = sorted(gamma_values, reverse=bool(_descending))
    if verbose:
        print(f"Tolerance : {tol}, Perturbation constant : {perturb_const}")
    for mode in modes:
        if verbose:
            print(f"\n===== Training for mode {mode} =====")
        mu_logs, models_by_gamma, history_by_gamma, epochs_by_gamma = [], {}, {}, {}
        prev_model = None
        normal_const = None
        for gamma in gamma_values:
            if verbose:
                print(f"\nTraining for {_label} = {gamma:.2f}, mode = {mode}, nonlinearity p = {p}")
            model = _make_model(mode, gamma) if _make_model else (_cls or _RefinePINN)(layers, mode=mode, gamma=gamma, **model_kw)
            if prev_model is not None:
                model.load_state_dict(prev_model.state_dict())
            elif gamma == 0.0 and pretrain is not None:
                if pretrain == "reference":
                    model = _refine_pretrain(model, mode, X_train, epochs=2000, lr=1e-3, verbose=verbose)   # :302-303
                else:
                    model = pretrain(model, mode, X_train)
            else:
                model.apply(lambda m: init_fn(m, mode))
            if normal_const is None:                                   # :333-335
                nn0 = model.forward(X_dev)
                normal_const = float(nn0.max())
                constant_history[mode] = torch.tensor(normal_const)
            model.perturb_scale = perturb_const / normal_const
            model._pull()
            model.start_flat = model._flat.copy()                      # weights this stage started from (after pre-training / warm start)
            eng = model._get_engine(**pot_over, p=int(p), dx=float(dx), lr=float(lr),
                                    sched=capi.SCHED_COSINE_LOSS, T_0=200.0, T_mult=2.0, eta_min=1e-6,
                                    w_bc=10.0, w_norm=20.0, stop_tol=float(tol), stop_patience=2000,
                                    history_capacity=max(int(epochs), 1))
            model._bind_training_data(eng, X, X_dev, bpts)
            done = 0
            final_epoch = epochs
            step_n = chunk
            while done < epochs:
                n = min(step_n, epochs - done)
                eng.run(n)
                done += n
                stopped, stop_step = eng.stop_state()
                if stopped:
                    final_epoch = stop_step - 1                        # reference epochs are 0-based
                    if verbose:
                        print(f"Early stop at epoch {final_epoch}")
                    break
                # epochs enqueued behind the stop epoch run frozen (parameters untouched) but still cost their launches: look more
                # often once the loss is within a factor of four of the tolerance (the stop epoch itself stays exact either way)
                if tol > 0 and chunk > 32:
                    step_n = 32 if eng.read_scalars()["loss"] < 4.0 * tol else chunk
            n_rec = (final_epoch + 1) if final_epoch < epochs else epochs
            hist = _history(eng, 1, n_rec)
            loss_history = hist["loss"][::10].tolist()                                       # :375-376
            lambda_history = hist["mu"][::100].tolist()                                      # :379-380
            constraint_history = (10.0 * hist["bc"][::100] + 20.0 * hist["norm"][::100]).tolist()
            if verbose:
                for i in range(0, len(hist), 500):
                    h = hist[i]
                    print(f"Epoch {i}, μ: {h['mu']:.4f}\nTotal Loss: {h['loss']:.6f}, PDE residual: {h['pde']:.6f}, "
                          f"Constraints: {10.0 * h['bc'] + 20.0 * h['norm']:.6f}")
            model._pull()
            model.history = hist                                       # every epoch's record (the returned histories keep the reference's cadence)
            final_mu = lambda_history[-1] if lambda_history else 0                           # :407 (quirk Q5)
            mu_logs.append((gamma, final_mu))
            model.last_mu = float(hist["mu"][-1]) if len(hist) else float("nan")
            models_by_gamma[gamma] = model
            history_by_gamma[gamma] = {"loss": loss_history, "constraint": constraint_history, "lambda": lambda_history}
            epochs_by_gamma[gamma] = final_epoch
            if prev_model is not None and prev_model is not model:
                prev_model.close()
            prev_model = model
        if prev_model is not None:
            prev_model.close()
        mu_table[mode] = mu_logs
        models_by_mode[mode] = models_by_gamma
        training_history[mode] = history_by_gamma
        epochs_history[mode] = epochs_by_gamma
    return models_by_mode, mu_table, training_history, constant_history, epochs_history


def _refine_wavefunction(model, X_test, constant, perturb_const, mode=None):
    """plot_wavefunction's numerical part (refine/...:459-474): normalised u on the test grid, |u| for mode 0."""
    X = _as_np(X_test).astype(np.float64)
    dx = X[1, 0] - X[0, 0]
    mode = model.mode if mode is None else mode
    model.perturb_scale = float(perturb_const) / float(constant)
    eng = model._get_engine()
    eng.set_perturb_scale(model.perturb_scale)
    u, dens = eng.eval_density(torch.as_tensor(X.astype(np.float32), device=_device()), float(dx), abs_flag=(mode == 0))
    return u.cpu().numpy().ravel()


# ================================================================================================
# notebook flavour driver
# ================================================================================================
def _nb_advanced_initialization(m, mode):
    m._flat = _advanced_init(m.layers, mode, "xavier_uniform", m._spec())
    if m._engine is not None:
        m._engine.set_params(m._flat)


def _nb_train(gamma_values, powers, modes, X_train, lb, ub, layers, epochs, potential_type="harmonic", lr=1e-3,
              verbose=True, chunk=500):
    """train_gpe_model of Gross_Pitaevskii_1D_power_Test.ipynb c10: loop modes x powers at gamma = gamma_values[0]."""
    if potential_type not in _POT:
        raise ValueError(f"Unknown potential type: {potential_type}")
    gamma = gamma_values[0]                                            # c10:L34
    X = _as_np(X_train).astype(np.float64)
    dx = X[1, 0] - X[0, 0]
    dev = _device()
    X_dev = torch.as_tensor(X.astype(np.float32), device=dev)
    bpts = torch.tensor([[lb], [ub]], dtype=torch.float32, device=dev)
    models_by_mode, mu_table = {}, {}
    powers = sorted(powers)
    for mode in modes:
        if verbose:
            print(f"\n===== Training for mode {mode} =====")
        mu_logs, models_by_power, prev_model = [], {}, None
        for power in powers:
            if verbose:
                print(f"\nTraining for power = {power:.2f}, mode = {mode}")
            model = _NotebookPINN(layers, mode=mode, power=power)
            if prev_model is not None:
                model.load_state_dict(prev_model.state_dict())
            else:
                model.apply(lambda m: _nb_advanced_initialization(m, mode))
            model.gamma = gamma
            model._pull()
            model.start_flat = model._flat.copy()                      # weights this stage started from
            eng = model._get_engine(potential=_POT[potential_type], p=int(power), dx=float(dx), lr=float(lr),
                                    sched=capi.SCHED_PLATEAU, factor=0.5, patience=100, min_lr=1e-5,
                                    w_bc=10.0, w_norm=20.0, history_capacity=max(int(epochs), 1))
            eng.bind_points(X_dev)
            eng.bind_boundary(bpts)
            done = 0
            while done < epochs:
                n = min(chunk, epochs - done)
                eng.run(n)
                done += n
            hist = _history(eng, 1, epochs)
            lambda_history = hist["mu"][::100].tolist()                                      # c10:L106-108
            if verbose:
                for i in range(0, len(hist), 500):
                    h = hist[i]
                    print(f"Epoch {i}, Loss: {h['loss']:.6f}, μ: {h['mu']:.4f}")
            model._pull()
            model.history = hist
            final_mu = lambda_history[-1] if lambda_history else 0
            mu_logs.append((power, final_mu))
            models_by_power[power] = model
            if prev_model is not None:
                prev_model.close()
            prev_model = model
        if prev_model is not None:
            prev_model.close()
        mu_table[mode] = mu_logs
        models_by_mode[mode] = models_by_power
    return models_by_mode, mu_table


def _nb_density(model, X_test, mode=None):
    """plot_wavefunction_densities' numerical part (nb c12:L30-42): normalised |psi|^2 on the test grid."""
    X = _as_np(X_test).astype(np.float64)
    dx = X[1, 0] - X[0, 0]
    eng = model._get_engine()
    u, dens = eng.eval_density(torch.as_tensor(X.astype(np.float32), device=_device()), float(dx), abs_flag=False)
    return dens.cpu().numpy()


def _box_train(gamma_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const, potential_type="box", lr=1e-5,
               verbose=True, L=1.0, **kw):
    """train_gpe_model of refine/box_pinn_simulation.py:267-470 (same loop as the harmonic one, box model class)."""
    if potential_type != "box":
        raise ValueError(f"Unknown potential type: {potential_type}")
    return _refine_train(gamma_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const, "box", lr, verbose,
                         _cls=_BoxPINN, L=L, **kw)


def _gravity_train(gamma_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const, potential_type="gravity_well",
                   lr=1e-5, verbose=True, **kw):
    """train_gpe_model of refine/gravity_well_pinn_simulation.py (same loop; sum/sum lambda == mean/mean)."""
    if potential_type != "gravity_well":
        raise ValueError(f"Unknown potential type: {potential_type}")
    return _refine_train(gamma_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const, "gravity_well", lr,
                         verbose, _cls=_GravityWellPINN, **kw)


def _b2g_train(gamma_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const, potential_type="gaussian", lr=1e-5,
               verbose=True, **kw):
    """train_gpe_model of refine/box_to_gaussian_pinn_simulation.py:242-449 (the same loop; no pre-training there: :316-320)."""
    if potential_type != "gaussian":
        raise ValueError(f"Unknown potential type: {potential_type}")
    kw.setdefault("pretrain", None)
    return _refine_train(gamma_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const, "gaussian", lr, verbose,
                         _cls=_BoxToGaussianPINN, L=float(ub), **kw)


def _vbeta_advanced_initialization(m, mode):
    """advanced_initialization of the beta-sweep scripts (refine/vary_potential_parameter_harmonic.py:788-798): from mode 3 on a ten
    times smaller gain and bias 1e-4."""
    m._flat = _advanced_init(m.layers, mode, "xavier_normal_vbeta", m._spec())
    if m._engine is not None:
        m._engine.set_params(m._flat)


def _vbeta_train(cls, valid_potentials, with_L):
    """train_gpe_model(gamma, beta_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const, potential_type, lr, verbose)
    of refine/vary_potential_parameter_harmonic.py:344-556 (gravity well :259-470, box + Gaussian :227-439): the refine family's epoch
    body and early stopping, ONE interaction strength gamma, warm-started continuation in beta (sorted ascending), pre-training on the
    analytic base when the first beta is 0.0 (:427-430), normal_const taken at the first epoch of the first beta (the scripts define it
    only when that beta is 0 -- 1 in the gravity-well script, :374 there -- and raise NameError otherwise: quirk Q6)."""
    def train(gamma, beta_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const, potential_type="box", lr=1e-5,
              verbose=True, **kw):
        if potential_type not in valid_potentials:         # the scripts raise this from compute_potential in the first epoch body
            raise ValueError(f"Unknown potential type: {potential_type}")

        def make(mode, beta):
            mdl = cls(layers, mode=mode, beta=beta, L=ub) if with_L else cls(layers, mode=mode, beta=beta)
            mdl.gamma = float(gamma)
            mdl._pot_name = potential_type
            return mdl

        return _refine_train(beta_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const, potential_type, lr, verbose,
                             _make_model=make, _init_fn=_vbeta_advanced_initialization, _pot_over={}, _label="β", **kw)
    return train


vary_beta_harmonic = types.SimpleNamespace(GrossPitaevskiiPINN=_VaryHarmonicPINN, train_gpe_model=_vbeta_train(_VaryHarmonicPINN, ("harmonic",), True),
                                           advanced_initialization=_vbeta_advanced_initialization,
                                           pretrain_on_analytical_solution=_refine_pretrain, normalized_wavefunction=_refine_wavefunction)
vary_beta_gravity_well = types.SimpleNamespace(GrossPitaevskiiPINN=_VaryGravityPINN, train_gpe_model=_vbeta_train(_VaryGravityPINN, ("gravity_well",), False),
                                               advanced_initialization=_vbeta_advanced_initialization,
                                               pretrain_on_analytical_solution=_refine_pretrain)
vary_beta_box_and_gaussian = types.SimpleNamespace(GrossPitaevskiiPINN=_VaryBoxGaussianPINN,
                                                   train_gpe_model=_vbeta_train(_VaryBoxGaussianPINN, ("box", "gaussian"), True),
                                                   advanced_initialization=_vbeta_advanced_initialization,
                                                   pretrain_on_analytical_solution=_refine_pretrain)
box_to_gaussian = types.SimpleNamespace(GrossPitaevskiiPINN=_BoxToGaussianPINN, train_gpe_model=_b2g_train,
                                        advanced_initialization=_refine_advanced_initialization)
gravity_well = types.SimpleNamespace(GrossPitaevskiiPINN=_GravityWellPINN, train_gpe_model=_gravity_train,
                                     advanced_initialization=_refine_advanced_initialization,
                                     pretrain_on_analytical_solution=_refine_pretrain)
box = types.SimpleNamespace(GrossPitaevskiiPINN=_BoxPINN, train_gpe_model=_box_train,
                            advanced_initialization=_refine_advanced_initialization,
                            pretrain_on_analytical_solution=_refine_pretrain)
def _refine_negative_train(gamma_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const, potential_type="harmonic",
                           lr=1e-5, verbose=True, **kw):
    """train_gpe_model of refine/harmonic_pinn_simulation_negative_interaction_strength.py:217-428: the same residual and epoch body as
    the main script, the continuation walks the (sorted) interaction strengths DOWNWARDS -- 0, -0.5, ... (:271, :286) -- so the
    warm start and normal_const come from gamma = 0 although the attractive values sort first."""
    return _refine_train(gamma_values, modes, p, X_train, lb, ub, layers, epochs, tol, perturb_const, potential_type, lr, verbose,
                         _descending=True, **kw)


refine_negative = types.SimpleNamespace(GrossPitaevskiiPINN=_RefinePINN, train_gpe_model=_refine_negative_train,
                                        advanced_initialization=_refine_advanced_initialization,
                                        pretrain_on_analytical_solution=_refine_pretrain,
                                        normalized_wavefunction=_refine_wavefunction, KIND="xavier_normal")
refine = types.SimpleNamespace(GrossPitaevskiiPINN=_RefinePINN, train_gpe_model=_refine_train,
                               advanced_initialization=_refine_advanced_initialization,
                               pretrain_on_analytical_solution=_refine_pretrain,
                               normalized_wavefunction=_refine_wavefunction, KIND="xavier_normal")
notebook = types.SimpleNamespace(GrossPitaevskiiPINN=_NotebookPINN, train_gpe_model=_nb_train,
                                 advanced_initialization=_nb_advanced_initialization, density=_nb_density,
                                 KIND="xavier_uniform")


# ================================================================================================
# 2D classes: src/gross_pitaevskii_2D.py (class :16-274) and src/gross_pitaevskii_2D_minimal.py (class :12-222, train_pinn :278-327)
# ================================================================================================
_PINN2D_FLAVOR = _Flavor("pinn2d", capi.ACT_TANH, 1.0, 1.0, 0, 0.0, "xavier_uniform")


class _PINN2D(_PINNBase):
    """GrossPitaevskiiPINN(layers, hbar, m, g) of the two 2D scripts: -lap u + V u + g |u|^2 u = lambda u on the disk around (pi/2, pi/2),
    Gaussian potential, plain tanh network, no analytic base.  Their loss (src/gross_pitaevskii_2D.py:215-242):
        10 mean(u(x_bc)^2)  +  1/2 (sum |grad u|^2 + sum V u^2 + g/2 sum u^4)  +  mean(r^2) + 1/(mean u^2 + 1e-2) + 1/(lambda^2 + 1e-6),
        lambda = mean(|grad u|^2 + V u^2 + g u^4) / mean(u^2),   r = -lap u + V u + g |u^2| u - lambda u.
    Quirk Q1: the reference multiplies V [N] by u [N,1], so for N > 1 its lambda and residual are [N,N] broadcasts; this class
    evaluates the per-point formulas above, which is what the reference computes when called with one point (the golden vectors)."""
    _flavor = _PINN2D_FLAVOR

    def __init__(self, layers, hbar=1.0, m=1.0, g=100.0):
        super().__init__(layers, hbar, m, mode=0, gamma=g)
        self.use_perturbation = False

    @property
    def g(self):
        return self.gamma

    @g.setter
    def g(self, v):
        self.gamma = v
        if self._engine is not None:
            self._engine.set_gamma(float(v))

    _W_FULL = (1.0, 10.0, 0.0, 0.0, 0.0, 1.0)         # w_pde, w_bc (boundary_loss's own x 10), w_norm, w_sym, w_orth, w_riesz

    def _extra_config(self):
        return dict(potential=capi.POT_PRECOMPUTED, p=3, abs_power=True, w_pde=1.0, w_bc=10.0, w_norm=0.0, w_riesz=1.0,
                    riesz_kind=capi.RIESZ_SUM, lambda_kind=capi.LAMBDA_ENERGY, w_reg_f=1.0, reg_f_eps=1e-2, w_reg_lam=1.0,
                    reg_lam_eps=1e-6, clip_norm=0.0, lr=1e-3, dx=1.0)

    def compute_potential(self, inputs, V0=1.0, x0=np.pi / 2, y0=np.pi / 2, sigma=0.5):
        """V0 exp(-((x - x0)^2 + (y - y0)^2) / (2 sigma^2))   (src/gross_pitaevskii_2D.py:244-274) -> [N]"""
        x, y = inputs[:, 0], inputs[:, 1]
        return V0 * torch.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))

    def _bind(self, eng, inputs, boundary=None):
        x = eng._to_dev(inputs.detach() if isinstance(inputs, torch.Tensor) else inputs, "x")
        eng.bind_points(x, self.compute_potential(x))
        eng.bind_boundary(boundary)
        return x

    def _terms(self, inputs, predictions=None, boundary=None, want_fields=False):
        eng = self._get_engine()
        eng.set_gamma(float(self.gamma))
        eng.set_loss_weights(*(self._W_FULL if boundary is not None else (1.0, 0.0, 0.0, 0.0, 0.0, 1.0)))
        eng.set_perturb_scale(1.0 if predictions is None else self._scale_of(inputs, predictions))
        self._bind(eng, inputs, boundary)
        out = eng.residual(want_fields=want_fields)
        eng.set_loss_weights(*self._W_FULL)
        return out

    def pde_loss(self, inputs, predictions):
        """-> (mean(r^2) + L_f + L_lambda, r [N,1], lambda)   (src/gross_pitaevskii_2D.py:154-213)"""
        sc, _, res = self._terms(inputs, predictions, want_fields=True)
        dev = _device()
        return (torch.tensor(sc["pde"] + sc["reg"], dtype=torch.float32, device=dev), res,
                torch.tensor(sc["mu"], dtype=torch.float32, device=dev))

    def riesz_loss(self, predictions, inputs):
        """1/2 (sum |grad u|^2 + sum V u^2 + g/2 sum u^4)   (src/gross_pitaevskii_2D.py:112-151)"""
        sc, _, _ = self._terms(inputs, predictions)
        return torch.tensor(sc["riesz"], dtype=torch.float32, device=_device())

    def boundary_loss(self, x_bc, y_bc):
        """10 mean(u(x_bc)^2): the reference replaces y_bc by zeros (src/gross_pitaevskii_2D.py:83-109)"""
        eng = self._get_engine()
        eng.set_loss_weights(*self._W_FULL)
        return 10.0 * super().boundary_loss(x_bc, None)

    def loss(self, x, x_bc, u_bc):
        """boundary_loss + pde_loss + riesz_loss, weights 1 (src/gross_pitaevskii_2D.py:215-242)"""
        sc, _, _ = self._terms(x, None, boundary=x_bc)
        return torch.tensor(sc["loss"], dtype=torch.float32, device=_device())

    total_loss = loss                                  # the name in src/gross_pitaevskii_2D_minimal.py:201-222


def _pinn2d_initialize_weights(m):
    """initialize_weights (src/gross_pitaevskii_2D.py:614-618): Xavier-uniform weights, bias 0.01, from torch's CPU RNG in module order."""
    m._flat = _advanced_init(m.layers, 0, "xavier_uniform", m._spec())
    if m._engine is not None:
        m._engine.set_params(m._flat)


def _pinn2d_boundary(N_u, center, radius):
    theta = np.linspace(0, 2 * np.pi, N_u)
    X_u = np.column_stack((center[0] + radius * np.cos(theta), center[1] + radius * np.sin(theta)))
    return X_u, np.zeros((X_u.shape[0], 1))


def _pinn2d_prepare_polar(N_u, N_f, center=(np.pi / 2, np.pi / 2), radius=np.pi / 2):
    """prepare_training_data of src/gross_pitaevskii_2D.py:277-295: N_u points on the circle; N_f collocation points drawn as (angle,
    radius) pairs from numpy's global generator -- the reference draws them one pair at a time, this draws the same stream as one block."""
    X_u, u = _pinn2d_boundary(N_u, center, radius)
    r = np.random.random_sample((N_f, 2))
    ang, rad = 0.0 + (2 * np.pi - 0.0) * r[:, 0], 0.0 + (radius - 0.0) * r[:, 1]
    X_f = np.column_stack((center[0] + rad * np.cos(ang), center[1] + rad * np.sin(ang)))
    return X_f, X_u, u


def _pinn2d_prepare_square(N_u, N_f, center=(np.pi / 2, np.pi / 2), radius=np.pi / 2):
    """prepare_training_data of src/gross_pitaevskii_2D_minimal.py:225-261: uniform draws in the bounding square, kept inside the disk
    (so fewer than N_f points come back)."""
    X_u, u = _pinn2d_boundary(N_u, center, radius)
    cx = np.random.uniform(center[0] - radius, center[0] + radius, N_f)
    cy = np.random.uniform(center[1] - radius, center[1] + radius, N_f)
    keep = (cx - center[0]) ** 2 + (cy - center[1]) ** 2 <= radius ** 2
    return np.column_stack((cx[keep], cy[keep])), X_u, u


def _pinn2d_solution_on_grid(model, num_grid_pts=100, center=(np.pi / 2, np.pi / 2), radius=np.pi / 2):
    """numerical part of plot_solution (src/gross_pitaevskii_2D_minimal.py:330-371): the prediction on the plotting grid -> X, Y, u [n,n]"""
    xv = np.linspace(center[0] - radius, center[0] + radius, num_grid_pts)
    yv = np.linspace(center[1] - radius, center[1] + radius, num_grid_pts)
    X, Y = np.meshgrid(xv, yv)
    pts = np.hstack((X.flatten()[:, None], Y.flatten()[:, None])).astype(np.float32)
    u = model.forward(torch.as_tensor(pts, device=_device())).cpu().numpy().reshape(num_grid_pts, num_grid_pts)
    return X, Y, u


def _pinn2d_train(N_u=500, N_f=10000, layers=[2, 400, 400, 400, 1], epochs=1000, g=100.0, verbose=True, chunk=400, data=None,
                  prepare=_pinn2d_prepare_square):
    """train_pinn of src/gross_pitaevskii_2D_minimal.py:278-327: model + initialize_weights + Adam(lr 1e-3), prepare_training_data,
    `epochs` full-batch steps on total_loss, a progress line every 400 epochs (the reference also draws a figure there: plot_solution's
    numbers are solution_on_grid).  Extras: `g` (the reference builds the class with its default 100), `data` = (X_f, X_u, u) to skip
    the random draw.  Returns the model; model.history holds every epoch's record (loss, mu, pde, bc, riesz, reg, ...)."""
    model = _PINN2D(layers, g=g)
    model.apply(_pinn2d_initialize_weights)
    X_f, X_u, u_train = data if data is not None else prepare(N_u, N_f)
    dev = _device()
    model.start_flat = model._flat.copy()
    eng = model._get_engine(history_capacity=max(int(epochs), 1))
    eng.set_loss_weights(*model._W_FULL)
    model._bind(eng, torch.as_tensor(np.asarray(X_f, dtype=np.float32), device=dev),
                torch.as_tensor(np.asarray(X_u, dtype=np.float32), device=dev))
    done = 0
    while done < epochs:
        n = min(chunk, epochs - done)
        eng.run(n)
        done += n
    hist = _history(eng, 1, epochs) if epochs > 0 else _History([])
    if verbose:
        for i in range(0, len(hist), 400):
            print(f"Epoch [{i}/{epochs}], Loss: {hist[i]['loss']:.6f}")
    model._pull()
    model.history = hist
    return model


pinn2d = types.SimpleNamespace(GrossPitaevskiiPINN=_PINN2D, prepare_training_data=_pinn2d_prepare_polar, initialize_weights=_pinn2d_initialize_weights,
                               train_pinn=lambda *a, **k: _pinn2d_train(*a, **dict(dict(prepare=_pinn2d_prepare_polar), **k)),
                               solution_on_grid=_pinn2d_solution_on_grid)
pinn2d_minimal = types.SimpleNamespace(GrossPitaevskiiPINN=_PINN2D, prepare_training_data=_pinn2d_prepare_square,
                                       initialize_weights=_pinn2d_initialize_weights, train_pinn=_pinn2d_train,
                                       solution_on_grid=_pinn2d_solution_on_grid)
