"""Import shim: ``import mon_amd`` loads the package that lives in the directory
``etss-07-imageenhance_amd/`` (its name is not a valid Python identifier)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "etss-07-imageenhance_amd")
_spec = _ilu.spec_from_file_location("mon_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["mon_amd"] = _mod
_spec.loader.exec_module(_mod)
