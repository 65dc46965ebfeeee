"""Import helper: the package directory is named ``firstorderlp.jl_amd`` (after
the reference repo), which is not a valid Python identifier, so it is loaded
under the module name ``firstorderlp_jl_amd``.

    import folp_loader; folp = folp_loader.load()
    from firstorderlp_jl_amd import optimize, PdhgParameters   # now importable
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "firstorderlp.jl_amd")
_NAME = "firstorderlp_jl_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_PKG_DIR, "__init__.py"),
        submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
