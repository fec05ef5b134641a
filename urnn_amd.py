"""Import shim: the package lives in ``./u-rnn_amd`` (not a valid Python identifier), this module
re-exports it under the importable name ``urnn_amd``."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "u-rnn_amd")
_spec = importlib.util.spec_from_file_location(
    "urnn_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["urnn_amd"] = _mod
_spec.loader.exec_module(_mod)
