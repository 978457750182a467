"""Import shim: the package directory is named `godot-volumetric-cloud-demo-v2_amd/` (not a valid Python
identifier), so this module loads it under the importable name `gvcd_amd`.  `import gvcd_amd` anywhere in
the repo (tests, bench.py, __graft_entry__.py) gives the package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "godot-volumetric-cloud-demo-v2_amd")
_spec = importlib.util.spec_from_file_location("gvcd_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gvcd_amd"] = _mod
_spec.loader.exec_module(_mod)
