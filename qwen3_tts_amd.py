"""Import alias: `import qwen3_tts_amd` (and `qwen3_tts_amd.<submodule>`) load the package in the directory `qwen3-tts_amd/`.
Every submodule exists ONCE, under its real name `qwen3-tts_amd.<submodule>`; the alias names resolve to those same module
objects, so classes and module state (the loaded library, staging buffers) are never duplicated."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_ALIAS, _REAL = __name__, "qwen3-tts_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if not name.startswith(_ALIAS + "."):
            return None
        try:
            importlib.import_module(_REAL + name[len(_ALIAS):])
        except ModuleNotFoundError as e:
            if e.name and e.name.startswith(_REAL):
                return None
            raise
        return importlib.util.spec_from_loader(name, self)

    def create_module(self, spec):
        return sys.modules[_REAL + spec.name[len(_ALIAS):]]

    def exec_module(self, module):
        pass

    def get_code(self, name):
        """`python -m qwen3_tts_amd.<submodule>`: runpy runs the real module's code as __main__."""
        real = _REAL + name[len(_ALIAS):]
        return sys.modules[real].__spec__.loader.get_code(real)


if not any(isinstance(f, _AliasFinder) or type(f).__name__ == "_AliasFinder" for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[_ALIAS] = _pkg
