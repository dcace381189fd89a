"""Import alias: ``import voxcarve`` loads the package directory
``voxel-based-3d-reconstruction_amd/`` (its name is not a valid Python identifier)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "voxel-based-3d-reconstruction_amd")
_spec = importlib.util.spec_from_file_location(
    "voxcarve", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["voxcarve"] = _mod
_spec.loader.exec_module(_mod)
