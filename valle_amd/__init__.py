"""Import alias: the product package lives in ``vall-e_amd/`` (the directory name the
build contract fixes), which is not a valid Python identifier.  ``import valle_amd``
loads that directory as the package ``valle_amd``."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "vall-e_amd")
_spec = _ilu.spec_from_file_location(
    "valle_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules["valle_amd"] = _mod
_spec.loader.exec_module(_mod)
