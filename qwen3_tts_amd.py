"""Import alias: `import qwen3_tts_amd` loads the package in the directory `qwen3-tts_amd/`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("qwen3-tts_amd")
sys.modules[__name__] = _pkg
