"""Host-side handle on the HIP engine (libgpe_hip.so).  PyTorch-ROCm tensors are used as device storage
and for torch.distributed (RCCL) only -- every number on the hot path is produced by the HIP kernels.

Replaces, for the hot path: the model/optimizer/scheduler objects and the epoch body of
refine/harmonic_pinn_simulation.py:295-361 and Gross_Pitaevskii_1D_power_Test.ipynb c10:L63-103.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np
import torch

from . import _capi as capi


class GPEError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[gpe status {code}] {msg}")
        self.code = code


@dataclass
class GPEConfig:
    """Python twin of include/gpe_hip.h:gpe_config (see the header for the reference citation of every field)."""
    layers: Sequence[int]
    activation: int = capi.ACT_TANH
    complex_psi: bool = False
    kinetic_coeff: float = 0.5
    potential: int = capi.POT_HARMONIC
    pot_scale: float = 0.5
    omega: Sequence[float] = (1.0, 1.0, 1.0)
    pot_a: float = 0.0
    pot_v0: float = 1.0
    pot_k: float = 2.0 * np.pi / 5.0
    omega_rot: float = 0.0
    gamma: float = 0.0
    p: int = 3
    abs_power: bool = False
    base_mode: int = -1
    base_deriv: int = 0
    perturb_scale: float = 1.0
    bc_nn_scale: float = 1.0
    w_pde: float = 1.0
    w_bc: float = 10.0
    w_norm: float = 20.0
    w_sym: float = 0.0
    w_orth: float = 0.0
    sym_sign: float = 1.0
    dx: float = 1.0
    n_global: int = 0
    lr: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    clip_norm: float = 1.0
    sched: int = capi.SCHED_CONST
    T_0: float = 200.0
    T_mult: float = 2.0
    eta_min: float = 1e-6
    factor: float = 0.5
    patience: int = 100
    min_lr: float = 1e-5
    threshold: float = 1e-4
    path: int = capi.PATH_AUTO
    world_size: int = 1
    history_capacity: int = 0
    stop_tol: float = 0.0
    stop_patience: int = 0
    base_kind: int = capi.BASE_HERMITE
    envelope: int = capi.ENV_NONE
    box_L: float = 1.0
    env_L: float = 1.0
    w_riesz: float = 0.0
    riesz_kind: int = capi.RIESZ_PAPER
    net_kind: int = capi.NET_MLP
    lambda_kind: int = capi.LAMBDA_RAYLEIGH      # LAMBDA_ENERGY: src/gross_pitaevskii_2D.py:192
    w_reg_f: float = 0.0                         # w / (mean u^2 + reg_f_eps)        src/gross_pitaevskii_2D.py:201
    reg_f_eps: float = 1e-2
    w_reg_lam: float = 0.0                       # w / (lambda^2 + reg_lam_eps)      src/gross_pitaevskii_2D.py:204
    reg_lam_eps: float = 1e-6

    def to_c(self) -> capi.gpe_config:
        c = capi.gpe_config()
        c.abi_version = capi.GPE_ABI_VERSION
        layers = [int(v) for v in self.layers]
        if len(layers) > capi.GPE_MAX_LAYERS:
            raise ValueError(f"at most {capi.GPE_MAX_LAYERS} layer entries")
        c.n_layers = len(layers)
        for i, v in enumerate(layers):
            c.layers[i] = v
        om = list(self.omega) + [1.0] * 3
        for i in range(3):
            c.omega[i] = float(om[i])
        for name in ("activation", "potential", "p", "base_mode", "base_deriv", "sched", "patience", "path",
                     "world_size", "history_capacity", "n_global", "stop_patience", "base_kind", "envelope", "riesz_kind", "net_kind", "lambda_kind"):
            setattr(c, name, int(getattr(self, name)))
        c.complex_psi = int(bool(self.complex_psi))
        c.abs_power = int(bool(self.abs_power))
        for name in ("kinetic_coeff", "pot_scale", "pot_a", "pot_v0", "pot_k", "omega_rot", "gamma", "perturb_scale",
                     "bc_nn_scale", "w_pde", "w_bc", "w_norm", "w_sym", "w_orth", "sym_sign", "dx", "lr", "beta1",
                     "beta2", "eps", "clip_norm", "T_0", "T_mult", "eta_min", "factor", "min_lr", "threshold",
                     "stop_tol", "box_L", "env_L", "w_riesz", "w_reg_f", "reg_f_eps", "w_reg_lam", "reg_lam_eps"):
            setattr(c, name, float(getattr(self, name)))
        return c

    @property
    def dim(self):
        return int(self.layers[0])

    @property
    def n_out(self):
        return int(self.layers[-1])

    @property
    def n_channels(self):
        return 1 + 2 * self.dim


def _check_dev_f32(t: torch.Tensor, name: str):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous float32 tensor on the GPU")


def _store_get(store, key, timeout, rank):
    """ncclUniqueId from a key-value store, bounded in time: a rank whose key never appears (ranks that created a different number
    of communicators look up different default keys) must fail with the key's name instead of blocking for ever."""
    import time
    if hasattr(store, "wait"):                           # torch.distributed stores: a wait with its own timeout
        import datetime
        try:
            store.wait([key], datetime.timedelta(seconds=timeout))
        except Exception as ex:
            raise TimeoutError(f"comm_init: rank {rank} found no ncclUniqueId under store key {key!r} within {timeout:.0f} s "
                               f"(does every rank create its communicators in the same order / publish under this key?)") from ex
        return bytes(store.get(key))
    deadline = time.monotonic() + timeout
    while True:
        try:
            v = store.get(key)
        except KeyError:
            v = None
        if v is not None:
            return bytes(v)
        if time.monotonic() > deadline:
            raise TimeoutError(f"comm_init: rank {rank} found no ncclUniqueId under store key {key!r} within {timeout:.0f} s")
        time.sleep(0.01)


class Engine:
    """One engine per process per GPU.  The HIP extension is mandatory (no fallback)."""

    def __init__(self, cfg: GPEConfig, device: Optional[int] = None, use_torch_stream: bool = True):
        if not torch.cuda.is_available():
            raise GPEError(capi.GPE_ERR_HIP, "no GPU visible: the GPE engine has no CPU fallback")
        self.lib = capi.load()
        self.cfg = cfg
        self.device = torch.cuda.current_device() if device is None else int(device)
        torch.cuda.set_device(self.device)
        # kernels are enqueued on torch's current stream so that they are ordered with RCCL collectives
        stream = torch.cuda.current_stream(self.device).cuda_stream if use_torch_stream else 0
        self._h = C.c_void_p()
        ccfg = cfg.to_c()
        rc = self.lib.gpe_create(C.byref(ccfg), self.device, C.c_void_p(stream), C.byref(self._h))
        if rc != capi.GPE_OK:
            msg = self.lib.gpe_last_error(None)
            code = capi.GPE_ERR_INVALID if rc == capi.GPE_ERR_INVALID else rc
            err = msg.decode() if msg else "gpe_create failed"
            if rc == capi.GPE_ERR_INVALID:
                raise ValueError(err)          # the reference raises ValueError for bad problem descriptions
            raise GPEError(code, err)
        self.n_params = int(self.lib.gpe_param_count(self._h))
        self._keep = {}
        # caller-owned exchange buffers (so torch.distributed can all-reduce them in place)
        nd = int(self.lib.gpe_exchange_dbl_count())
        self._xdbl = torch.zeros(nd, dtype=torch.float64, device=f"cuda:{self.device}")
        gp, gn = C.c_void_p(), C.c_int64()                  # (length of the engine's gradient message: the network as it RUNS -- hidden
        self._chk(self.lib.gpe_exchange_grad(self._h, C.byref(gp), C.byref(gn)))     # widths without a kernel instance are zero-padded -- + 4)
        self._xgrad = torch.zeros(int(gn.value), dtype=torch.float32, device=f"cuda:{self.device}")
        self._chk(self.lib.gpe_use_external_exchange(self._h, C.c_void_p(self._xdbl.data_ptr()), nd,
                                                     C.c_void_p(self._xgrad.data_ptr()), self._xgrad.numel()))
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.lib.gpe_exchange_sums(self._h, C.byref(p), C.byref(n)))
        assert p.value == self._xdbl.data_ptr()
        self._sums_view = self._xdbl[: n.value]

    # ---- plumbing -----------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != capi.GPE_OK:
            msg = self.lib.gpe_last_error(self._h)
            raise GPEError(rc, msg.decode() if msg else "?")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.gpe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def active_path(self) -> int:
        return int(self.lib.gpe_active_path(self._h))

    @property
    def active_kernels(self) -> dict:
        """{'fwd': name, 'bwd': name} of the jet forward / reverse kernels the bound collocation batch is dispatched to."""
        buf = C.create_string_buffer(512)
        self._chk(self.lib.gpe_active_kernels(self._h, buf, 512))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";"))

    # ---- data-parallel exchange inside the engine (RCCL on the engine's exchange stream) ---------------------------------
    _comm_seq = 0          # communicators created by this process: every rank creates them in the same order

    def comm_init(self, rank: int, world: int, store=None, key: str = None, timeout: float = 300.0):
        """Create the engine's own RCCL communicator.  The 128-byte ncclUniqueId of rank 0 reaches the other ranks through
        `store` (anything with set / get) or, when none is given, by torch.distributed.broadcast_object_list over the initialised
        default group.  `key`: the FULL store key, used as given (non-Python ranks can publish / read it under the same name); left
        at None it is "gpe_comm_id/<n>", n = communicators this process has created so far -- every rank must then create its
        communicators in the same order.  A rank that does not find the id within `timeout` seconds raises, naming the key."""
        if key is None:
            key = f"gpe_comm_id/{Engine._comm_seq}"
        Engine._comm_seq += 1
        ident = (C.c_ubyte * capi.GPE_COMM_ID_BYTES)()
        id_err = None
        if rank == 0:
            try:
                self._chk(self.lib.gpe_comm_unique_id(self._h, ident))
            except GPEError as ex:                         # (e.g. librccl not found) -- the other ranks are told below instead of left waiting
                if store is not None or world <= 1:
                    raise
                id_err = ex
        if store is not None:                              # caller's key-value store (set / get)
            if rank == 0:
                store.set(key, bytes(ident))
            else:
                C.memmove(ident, _store_get(store, key, timeout, rank), capi.GPE_COMM_ID_BYTES)
        elif world > 1:                                    # default: the initialised torch.distributed group, public API only
            import torch.distributed as dist
            box = [(bytes(ident) if id_err is None else None) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            if id_err is not None:
                raise id_err
            if box[0] is None:
                raise GPEError(capi.GPE_ERR_STATE, "comm_init: rank 0 could not create an ncclUniqueId")
            if rank != 0:
                C.memmove(ident, box[0], capi.GPE_COMM_ID_BYTES)
        self._chk(self.lib.gpe_comm_init(self._h, ident, int(rank), int(world)))

    def comm_info(self) -> dict:
        r, w, n = C.c_int(), C.c_int(), C.c_int64()
        self._chk(self.lib.gpe_comm_info(self._h, C.byref(r), C.byref(w), C.byref(n)))
        return dict(rank=r.value, world=w.value, collectives=n.value)

    def comm_destroy(self):
        self._chk(self.lib.gpe_comm_destroy(self._h))

    def comm_set_async(self, on: bool = True):
        """OPT-IN one-step-stale gradient (gradient all-reduce overlapped with the next forward); changes the trajectory."""
        self._chk(self.lib.gpe_comm_set_async(self._h, int(on)))

    def step_dp(self):
        """One data-parallel step with both exchanges issued by the engine (no Python between the phases, no host sync)."""
        self._chk(self.lib.gpe_step_dp(self._h))

    def run_dp(self, n_steps: int):
        self._chk(self.lib.gpe_run_dp(self._h, int(n_steps)))

    # ---- parameters ---------------------------------------------------------------------------------
    def set_params(self, flat):
        a = np.ascontiguousarray(np.asarray(flat, dtype=np.float32).ravel())
        self._chk(self.lib.gpe_set_params(self._h, a.ctypes.data_as(C.c_void_p), a.size))

    def get_params(self) -> np.ndarray:
        a = np.empty(self.n_params, dtype=np.float32)
        self._chk(self.lib.gpe_get_params(self._h, a.ctypes.data_as(C.c_void_p), a.size))
        return a

    def get_grad(self) -> np.ndarray:
        a = np.empty(self.n_params, dtype=np.float32)
        self._chk(self.lib.gpe_get_grad(self._h, a.ctypes.data_as(C.c_void_p), a.size))
        return a

    def get_adam_state(self):
        m = np.empty(self.n_params, dtype=np.float32)
        v = np.empty(self.n_params, dtype=np.float32)
        step = C.c_int64()
        self._chk(self.lib.gpe_get_adam_state(self._h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
                                              m.size, C.byref(step)))
        return m, v, int(step.value)

    def set_adam_state(self, m, v, step):
        m = np.ascontiguousarray(m, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        self._chk(self.lib.gpe_set_adam_state(self._h, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
                                              m.size, int(step)))

    def reset_optimizer(self, lr: float):
        self._chk(self.lib.gpe_reset_optimizer(self._h, float(lr)))

    # ---- data ---------------------------------------------------------------------------------------------
    def _to_dev(self, a, name):
        if isinstance(a, torch.Tensor):
            t = a.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous()
        else:
            t = torch.as_tensor(np.asarray(a, dtype=np.float32), device=f"cuda:{self.device}").contiguous()
        _check_dev_f32(t, name)
        return t

    def bind_points(self, x, V=None):
        x = self._to_dev(x, "x")
        if x.dim() != 2 or x.shape[1] != self.cfg.dim:
            raise ValueError(f"x must be [N,{self.cfg.dim}]")
        Vt = None if V is None else self._to_dev(V, "V").reshape(-1)
        self._keep["x"], self._keep["V"] = x, Vt
        self._chk(self.lib.gpe_bind_points(self._h, C.c_void_p(x.data_ptr()), x.shape[0],
                                           C.c_void_p(Vt.data_ptr()) if Vt is not None else None))
        self.n_local = int(x.shape[0])

    def bind_boundary(self, xb, target=None):
        if xb is None:
            self._chk(self.lib.gpe_bind_boundary(self._h, None, 0, None))
            return
        xb = self._to_dev(xb, "xb")
        tg = None if target is None else self._to_dev(target, "target")
        self._keep["xb"], self._keep["tg"] = xb, tg
        self._chk(self.lib.gpe_bind_boundary(self._h, C.c_void_p(xb.data_ptr()), xb.shape[0],
                                             C.c_void_p(tg.data_ptr()) if tg is not None else None))

    def bind_base(self, phi, phi1, phi2):
        """GPE_BASE_PRECOMPUTED: base function and its first two derivatives on the bound points."""
        ts = [self._to_dev(a, "base").reshape(-1) for a in (phi, phi1, phi2)]
        self._keep["base"] = ts
        self._chk(self.lib.gpe_bind_base(self._h, *[C.c_void_p(t.data_ptr()) for t in ts]))

    def bind_orth(self, k: int, psi_k):
        t = None if psi_k is None else self._to_dev(psi_k, "psi_k").reshape(-1)
        self._keep[f"orth{k}"] = t
        self._chk(self.lib.gpe_bind_orth(self._h, int(k), C.c_void_p(t.data_ptr()) if t is not None else None))

    # ---- forward-only -----------------------------------------------------------------------------------------
    def forward(self, x) -> torch.Tensor:
        x = self._to_dev(x, "x")
        out = torch.empty((x.shape[0], self.cfg.n_out), dtype=torch.float32, device=x.device)
        self._chk(self.lib.gpe_forward(self._h, C.c_void_p(x.data_ptr()), x.shape[0], C.c_void_p(out.data_ptr())))
        return out

    def forward_jets(self, x) -> torch.Tensor:
        x = self._to_dev(x, "x")
        out = torch.empty((self.cfg.n_channels, x.shape[0], self.cfg.n_out), dtype=torch.float32, device=x.device)
        self._chk(self.lib.gpe_forward_jets(self._h, C.c_void_p(x.data_ptr()), x.shape[0], C.c_void_p(out.data_ptr())))
        return out

    def residual(self, want_fields: bool = True):
        sc = capi.gpe_scalars()
        psi = res = None
        if want_fields:
            psi = torch.empty((self.n_local, self.cfg.n_out), dtype=torch.float32, device=f"cuda:{self.device}")
            res = torch.empty_like(psi)
        self._chk(self.lib.gpe_residual(self._h, C.byref(sc), C.c_void_p(psi.data_ptr()) if want_fields else None,
                                        C.c_void_p(res.data_ptr()) if want_fields else None))
        return sc.as_dict(), psi, res

    def eval_density(self, x, dx: float, abs_flag: bool = False):
        x = self._to_dev(x, "x")
        u = torch.empty((x.shape[0], self.cfg.n_out), dtype=torch.float32, device=x.device)
        dens = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
        self._chk(self.lib.gpe_eval_density(self._h, C.c_void_p(x.data_ptr()), x.shape[0], float(dx), int(abs_flag),
                                            C.c_void_p(u.data_ptr()), C.c_void_p(dens.data_ptr())))
        return u, dens

    # ---- training ---------------------------------------------------------------------------------------------------
    def step(self) -> dict:
        sc = capi.gpe_scalars()
        self._chk(self.lib.gpe_step(self._h, C.byref(sc)))
        return sc.as_dict()

    def run(self, n_steps: int):
        """n_steps steps enqueued back to back; no host synchronisation."""
        self._chk(self.lib.gpe_run(self._h, int(n_steps)))

    def step_begin(self):
        self._chk(self.lib.gpe_step_begin(self._h))

    def step_backward(self):
        self._chk(self.lib.gpe_step_backward(self._h))

    def step_update(self):
        self._chk(self.lib.gpe_step_update(self._h))

    @property
    def exchange_sums(self) -> torch.Tensor:
        """float64 device tensor (view) to all-reduce(sum) between step_begin and step_backward."""
        return self._sums_view

    @property
    def exchange_grad(self) -> torch.Tensor:
        """float32 device tensor [P+4] to all-reduce(sum) between step_backward and step_update."""
        return self._xgrad

    def step_distributed(self, group=None, sync: bool = False):
        """One data-parallel step: every rank holds a shard of the collocation points (dp.py)."""
        from .dp import distributed_step
        distributed_step(self, group)
        if sync:
            return self.read_scalars()
        return None

    def synchronize(self):
        self._chk(self.lib.gpe_synchronize(self._h))

    def stop_state(self):
        """(stopped, stop_step): early stopping evaluated on the device (refine/harmonic_pinn_simulation.py:389-400)."""
        s, st = C.c_int(), C.c_int64()
        self._chk(self.lib.gpe_stop_state(self._h, C.byref(s), C.byref(st)))
        return bool(s.value), int(st.value)

    def read_scalars(self) -> dict:
        sc = capi.gpe_scalars()
        self._chk(self.lib.gpe_read_scalars(self._h, C.byref(sc)))
        return sc.as_dict()

    def read_history(self, first_step: int, count: int):
        arr = (capi.gpe_scalars * count)()
        self._chk(self.lib.gpe_read_history(self._h, int(first_step), int(count), arr))
        return [a.as_dict() for a in arr]

    HISTORY_FIELDS = tuple(n for n, _ in capi.gpe_scalars._fields_)

    def read_history_array(self, first_step: int, count: int) -> np.ndarray:
        """The same records as one float64 array [count, len(HISTORY_FIELDS)] (no per-record Python objects: a 201-stage continuation
        reads 400 000 of them)."""
        arr = (capi.gpe_scalars * count)()
        self._chk(self.lib.gpe_read_history(self._h, int(first_step), int(count), arr))
        return np.frombuffer(arr, dtype=np.float64).reshape(count, len(self.HISTORY_FIELDS)).copy()

    # ---- pre-training on an analytic target (refine/harmonic_pinn_simulation.py:650-701) -----------------------------------
    def bind_target(self, target):
        t = None if target is None else self._to_dev(target, "target").reshape(-1, self.cfg.n_out)
        self._keep["target"] = t
        self._chk(self.lib.gpe_bind_target(self._h, C.c_void_p(t.data_ptr()) if t is not None else None))

    def mse_step(self) -> dict:
        sc = capi.gpe_scalars()
        self._chk(self.lib.gpe_mse_step(self._h, C.byref(sc)))
        return sc.as_dict()

    def mse_loss_grad(self):
        loss = C.c_double()
        self._chk(self.lib.gpe_mse_loss_grad(self._h, C.byref(loss)))
        return loss.value, self.get_grad()

    # ---- continuation knobs --------------------------------------------------------------------------------------------
    def set_gamma(self, g: float):
        self.cfg.gamma = float(g)
        self._chk(self.lib.gpe_set_gamma(self._h, float(g)))

    def set_power(self, p: int):
        self.cfg.p = int(p)
        self._chk(self.lib.gpe_set_power(self._h, int(p)))

    def set_lr(self, lr: float):
        self._chk(self.lib.gpe_set_lr(self._h, float(lr)))

    def set_perturb_scale(self, s: float):
        self.cfg.perturb_scale = float(s)
        self._chk(self.lib.gpe_set_perturb_scale(self._h, float(s)))

    def set_loss_weights(self, w_pde, w_bc, w_norm, w_sym=0.0, w_orth=0.0, w_riesz=0.0):
        """Change the loss weights between steps (what a host-side balancer such as ReLoBRaLo needs)."""
        w = (C.c_float * 6)(w_pde, w_bc, w_norm, w_sym, w_orth, w_riesz)
        self._chk(self.lib.gpe_set_loss_weights(self._h, w))
        self.cfg.w_pde, self.cfg.w_bc, self.cfg.w_norm = float(w_pde), float(w_bc), float(w_norm)
        self.cfg.w_sym, self.cfg.w_orth, self.cfg.w_riesz = float(w_sym), float(w_orth), float(w_riesz)

    def set_n_global(self, n: int):
        self.cfg.n_global = int(n)
        self._chk(self.lib.gpe_set_n_global(self._h, int(n)))

    def profile_enable(self, on: bool = True):
        self._chk(self.lib.gpe_profile_enable(self._h, int(on)))

    def profile_read(self):
        """-> dict(fwd_ms, fwd_launches, bwd_ms, bwd_launches) measured with HIP events on the engine stream."""
        out = (C.c_double * 4)()
        self._chk(self.lib.gpe_profile_read(self._h, out))
        return dict(fwd_ms=out[0], fwd_launches=int(out[1]), bwd_ms=out[2], bwd_launches=int(out[3]))

    def step_cost(self):
        f, b = C.c_double(), C.c_double()
        self._chk(self.lib.gpe_step_cost(self._h, C.byref(f), C.byref(b)))
        return f.value, b.value
