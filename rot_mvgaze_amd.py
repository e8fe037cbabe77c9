"""Import shim: the package directory is ``rot-mvgaze_amd/`` (not a legal Python identifier), so
``import rot_mvgaze_amd`` loads that directory as the package ``rot_mvgaze_amd``."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "rot-mvgaze_amd")
_spec = _u.spec_from_file_location("rot_mvgaze_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["rot_mvgaze_amd"] = _mod
_spec.loader.exec_module(_mod)
