"""Import shim: loads the package directory `kerneldensityestimate.jl_amd/` (whose name is not a
valid Python identifier) under the module name `kdehip`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kerneldensityestimate.jl_amd")
_spec = importlib.util.spec_from_file_location("kdehip", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["kdehip"] = _mod
_spec.loader.exec_module(_mod)
