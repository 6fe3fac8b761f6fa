"""Import shim: exposes the package directory ``pyflyt-drone_amd/`` (whose name
is not a valid Python identifier) as the module ``pyflyt_drone_amd``."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "pyflyt-drone_amd")
_spec = _u.spec_from_file_location("pyflyt_drone_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["pyflyt_drone_amd"] = _mod
_spec.loader.exec_module(_mod)
