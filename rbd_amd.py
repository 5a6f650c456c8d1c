"""Import shim: the product package lives in the directory `rigidbodydynamics.jl_amd/` (the name mandated for
this repo contains a dot, which Python's import statement cannot spell).  `import rbd_amd` loads that
directory as the package `rigidbodydynamics_jl_amd` and re-exports its public names."""
import importlib.util
import os
import sys

_PKG_NAME = "rigidbodydynamics_jl_amd"
_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rigidbodydynamics.jl_amd")

if _PKG_NAME not in sys.modules:
    _spec = importlib.util.spec_from_file_location(_PKG_NAME, os.path.join(_PKG_DIR, "__init__.py"),
                                                   submodule_search_locations=[_PKG_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_PKG_NAME] = _mod
    _spec.loader.exec_module(_mod)
_pkg = sys.modules[_PKG_NAME]
globals().update({k: getattr(_pkg, k) for k in dir(_pkg) if not k.startswith("__")})
package = _pkg
