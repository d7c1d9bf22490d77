"""Loader: exposes the package directory `linearmixingmodels.jl_amd/` (whose name is not a valid Python
identifier) as the importable module `lmm_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "linearmixingmodels.jl_amd")
_spec = importlib.util.spec_from_file_location("lmm_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lmm_amd"] = _mod
_spec.loader.exec_module(_mod)
