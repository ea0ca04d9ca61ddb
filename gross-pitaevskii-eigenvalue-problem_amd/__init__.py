"""MI355X-native engine for the Gross-Pitaevskii eigenvalue-residual PINN training step.

The directory name carries hyphens (it mirrors the reference repository's name); import it as

    import gpe_pinn                     # repo-root shim, or
    importlib.import_module("gross-pitaevskii-eigenvalue-problem_amd")

The HIP shared library (libgpe_hip.so, built by __graft_entry__.build()) is mandatory: nothing in this package
computes on the CPU.
"""
from . import _capi as capi
from ._capi import (ACT_TANH, ACT_TANH_PLUS1, POT_GAUSSIAN, POT_HARMONIC, POT_NONE, POT_PERIODIC, POT_PRECOMPUTED,
                    SCHED_CONST, SCHED_COSINE_LOSS, SCHED_PLATEAU, PATH_AUTO, PATH_GENERIC, PATH_FUSED)
from .engine import Engine, GPEConfig, GPEError
from . import dp, surface, checkpoint, relobralo
from .surface import refine, refine_negative, notebook, box, gravity_well, box_to_gaussian
from .surface import vary_beta_harmonic, vary_beta_gravity_well, vary_beta_box_and_gaussian
from .surface import pinn2d, pinn2d_minimal          # src/gross_pitaevskii_2D.py, src/gross_pitaevskii_2D_minimal.py

vary_beta = vary_beta_harmonic      # refine/vary_potential_parameter_harmonic.py, the flavour SURVEY 8(f3) cites

__all__ = ["capi", "Engine", "GPEConfig", "GPEError", "dp", "surface", "checkpoint", "refine", "refine_negative", "notebook", "box", "gravity_well", "box_to_gaussian",
           "vary_beta", "vary_beta_harmonic", "vary_beta_gravity_well", "vary_beta_box_and_gaussian", "pinn2d", "pinn2d_minimal"]
