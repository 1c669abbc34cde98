"""TEST INFRASTRUCTURE ONLY -- re-export of the synthetic weight / input generators.

The generators themselves live in the product package (`ppeadepth/synthetic.py`: bench.py and smoke() need random-init
weights and synthetic frames and must not import anything from oracle/); the golden generator and the tests reach
them through this module so that the reference model and this repo's model are filled by the same function.
"""
import os

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ppea-depth_amd", "ppeadepth")
import importlib.util as _ilu

_spec = _ilu.spec_from_file_location("_ppea_synthetic", os.path.join(_PKG, "synthetic.py"))
_mod = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(_mod)          # loaded by path: importing the package would require the HIP library
for _name in dir(_mod):
    if not _name.startswith("__"):
        globals()[_name] = getattr(_mod, _name)
