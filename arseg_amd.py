"""Import shim: the package directory is ``ar-seg_amd/`` (not a valid Python identifier), this
module loads it under the importable name ``arseg_amd`` and replaces itself in ``sys.modules``."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "ar-seg_amd")
_spec = _ilu.spec_from_file_location("arseg_amd", _os.path.join(_pkg_dir, "__init__.py"),
                                     submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["arseg_amd"] = _mod
_spec.loader.exec_module(_mod)
