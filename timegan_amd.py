"""Import shim: the package directory is named `time-gan_amd`, which is not a Python identifier.

`import timegan_amd` loads that directory as the package `timegan_amd`.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "time-gan_amd")
_spec = importlib.util.spec_from_file_location(
    "timegan_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["timegan_amd"] = _mod
_spec.loader.exec_module(_mod)
