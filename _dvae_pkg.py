"""Import helper: the package directory `d-vae_amd/` is not a valid Python
identifier, so register it in sys.modules under the name `dvae_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "d-vae_amd")


def _load():
    if "dvae_amd" in sys.modules:
        return sys.modules["dvae_amd"]
    spec = importlib.util.spec_from_file_location(
        "dvae_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["dvae_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


dvae_amd = _load()
