"""Import alias: ``import lwdetr_amd`` loads the package that lives in ``lw-detr_amd/``.

The directory name carries a hyphen (it mirrors the upstream project name), which is not a
legal Python identifier, so this one-file shim registers that directory as the package
``lwdetr_amd`` (sub-modules resolve inside ``lw-detr_amd/`` as usual).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lw-detr_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
