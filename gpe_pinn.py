"""Importable alias of the package directory `gross-pitaevskii-eigenvalue-problem_amd/` (hyphens are not valid in an
`import` statement).  `import gpe_pinn` gives the package itself."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("gross-pitaevskii-eigenvalue-problem_amd")
sys.modules[__name__] = _pkg
