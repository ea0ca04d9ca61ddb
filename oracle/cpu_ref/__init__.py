"""oracle/cpu_ref -- C++/OpenMP restatement of the jet algorithm (gpe_cpu_ref.cpp).  TEST INFRASTRUCTURE ONLY: the like-for-like
CPU baseline of bench.py and a second check of the jet algebra; the product never imports it.

build() compiles libgpe_cpu_ref.so next to the source (gcc -O3 -march=x86-64-v3 -fopenmp: AVX2 + FMA, safe on both the build
container and the GPU box's host).  sanitizer_selftest() builds selftest.cpp + the same source with
-fsanitize=address,undefined and runs it (CPU only -- GPU sanitizers are not available on this pool)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "gpe_cpu_ref.cpp")
LIB = os.path.join(HERE, "libgpe_cpu_ref.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        cmd = ["g++", "-O3", "-march=x86-64-v3", "-ffp-contract=fast", "-fopenmp", "-shared", "-fPIC", "-o", LIB, SRC]
        subprocess.check_call(cmd)
    return LIB


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        lib.cpu_ref_loss_grad.restype = C.c_int
        lib.cpu_ref_loss_grad.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                          C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int64, C.c_int,
                                          C.c_void_p, C.c_void_p]
        lib.cpu_ref_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def loss_grad(lib, pb, flat, x, threads: int = 0, want_grad: bool = True):
    """pb: oracle.gpe_oracle.Problem (real psi, harmonic potential, no base / boundary / symmetry terms are evaluated)."""
    layers = np.ascontiguousarray(np.asarray(pb.layers, np.int32))
    th = np.ascontiguousarray(np.asarray(flat, np.float32))
    xs = np.ascontiguousarray(np.asarray(x, np.float32))
    om = np.ascontiguousarray(np.asarray(list(pb.omega) + [1.0] * 3, np.float32)[:3])
    sc = np.zeros(8, np.float64)
    g = np.zeros(th.size, np.float32) if want_grad else None
    rc = lib.cpu_ref_loss_grad(layers.ctypes.data, layers.size, int(pb.activation == 1), th.ctypes.data, xs.ctypes.data, xs.shape[0],
                               float(pb.kinetic_coeff), float(pb.pot_scale), om.ctypes.data, float(pb.gamma), int(pb.p),
                               float(pb.w_pde), float(pb.w_norm), float(pb.dx), int(pb.n_global), int(threads), sc.ctypes.data,
                               g.ctypes.data if g is not None else None)
    if rc != 0:
        raise ValueError("cpu_ref: unsupported problem description (real psi, one output, dim <= 3, harmonic trap only)")
    names = ("loss", "pde", "norm", "mu", "num", "den", "sum_r2", "integral")
    return dict(zip(names, sc.tolist())), g


def step(lib, pb, flat, x, threads: int = 0):
    return loss_grad(lib, pb, flat, x, threads)


def sanitizer_selftest(timeout: int = 300) -> str:
    exe = os.path.join(HERE, "selftest_asan")
    cmd = ["g++", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fopenmp",
           "-o", exe, os.path.join(HERE, "selftest.cpp"), SRC]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", OMP_NUM_THREADS="4")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=timeout, env=env)
    if out.returncode != 0:
        raise RuntimeError("sanitizer self-test failed:\n" + out.stdout[-2000:] + out.stderr[-4000:])
    return out.stdout
