"""Import shim: the package directory is named ``wfl-asr_amd`` (not a valid Python
identifier), so ``import wfl_asr_amd`` lands here and re-binds the name to that package."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wfl-asr_amd")
_spec = importlib.util.spec_from_file_location(
    "wfl_asr_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["wfl_asr_amd"] = _mod
_spec.loader.exec_module(_mod)
