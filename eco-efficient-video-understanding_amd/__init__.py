"""MI355X-native ECO inference path (drop-in for the reference's pycaffe forward path).

Public surface mirrors ``caffe_3d/python/caffe/__init__.py`` / ``pycaffe.py`` for
the inference path only: ``Net``, ``TEST``/``TRAIN``, ``set_mode_gpu``,
``set_device``.  Import as ``import eco_amd as caffe`` (the directory name
``eco-efficient-video-understanding_amd`` is not a Python identifier; the
``eco_amd`` package at the repo root aliases it).
"""
from .netspec import TEST, TRAIN, NetSpec, NetSpecError  # noqa: F401
from .prototxt import parse as parse_prototxt  # noqa: F401

__all__ = ["TEST", "TRAIN", "NetSpec", "NetSpecError", "parse_prototxt",
           "Net", "Blob", "set_mode_gpu", "set_mode_cpu", "set_device"]


def __getattr__(name):
    # Lazy: the graph/parse layer must import without torch or the HIP library.
    if name in ("Net", "Blob", "set_mode_gpu", "set_mode_cpu", "set_device", "get_device"):
        from . import net as _net
        return getattr(_net, name)
    raise AttributeError(name)
