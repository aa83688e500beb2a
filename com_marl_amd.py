"""Import shim: the package directory is ``com-marl_amd/`` (hyphen, as the layout contract
names it), which Python cannot import by that spelling; this module loads it under the name
``com_marl_amd`` so that ``import com_marl_amd`` / ``from com_marl_amd import ...`` work."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.join(_here, "com-marl_amd")
_spec = importlib.util.spec_from_file_location("com_marl_amd", os.path.join(_pkg, "__init__.py"),
                                               submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["com_marl_amd"] = _mod
_spec.loader.exec_module(_mod)
