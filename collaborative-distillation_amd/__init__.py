"""collaborative-distillation_amd: MI355X-native WCT hot path.  The importable package is `wct_hip`
(this directory is put on sys.path); `importlib.import_module("collaborative-distillation_amd")`
also works and re-exports it."""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
if _here not in _sys.path:
    _sys.path.insert(0, _here)
import wct_hip  # noqa: E402,F401
from wct_hip import model_zoo  # noqa: E402,F401
