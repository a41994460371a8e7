"""Import helper: the package directory is named `gaussiansplatting.jl_amd` (with a
dot), so it is loaded by path and registered as `gaussiansplatting_jl_amd`."""
import importlib.util
import os
import sys

_NAME = "gaussiansplatting_jl_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gaussiansplatting.jl_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
