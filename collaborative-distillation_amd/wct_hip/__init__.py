"""wct_hip -- MI355X-native drop-in for the WCT stylisation path of Collaborative-Distillation.

`model_zoo` (layer graphs, weight blobs) imports without a GPU or torch; `WCT`, `styleTransfer`
need torch + a ROCm device and the in-tree libwct_hip.so.
"""
import importlib as _importlib

from . import model_zoo  # noqa: F401

__all__ = ["model_zoo", "WCT", "styleTransfer"]


def __getattr__(name):
    if name in ("WCT", "styleTransfer"):
        return getattr(_importlib.import_module(".wct", __name__), name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
