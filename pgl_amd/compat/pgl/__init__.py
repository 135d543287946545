"""`import pgl` -> pgl_amd.  Every `pgl.<x>` module name is answered with the `pgl_amd.<x>` module OBJECT (a meta-path
finder, not a second import of the files), so class identities, caches and the loaded libpglamd.so are shared whichever
name a script uses."""
import importlib
import importlib.abc
import importlib.machinery
import sys

import pgl_amd


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name == "pgl" or name.startswith("pgl."):
            return importlib.machinery.ModuleSpec(name, self)
        return None

    def create_module(self, spec):
        return importlib.import_module("pgl_amd" + spec.name[3:])

    def exec_module(self, module):
        return None


if not any(isinstance(f, _Alias) for f in sys.meta_path):
    sys.meta_path.insert(0, _Alias())
from pgl_amd import dataset as _dataset, graph_kernel as _gk          # noqa: E402,F401  (attributes the scripts reach as pgl.dataset ...)
from pgl_amd.utils import logger as _logger, data as _data             # noqa: E402,F401
sys.modules["pgl"] = pgl_amd
