"""Import shim: the package directory is `blocksparsematrices.jl_amd/` (a dot is not legal in a
Python module name), so `import bsm_amd` loads that directory as the package `bsm_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blocksparsematrices.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "bsm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bsm_amd"] = _mod
_spec.loader.exec_module(_mod)
