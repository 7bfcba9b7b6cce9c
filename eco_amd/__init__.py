"""Importable alias of ``eco-efficient-video-understanding_amd/`` (hyphens are not importable).

``import eco_amd as caffe`` gives the pycaffe-style surface of the MI355X ECO path.
"""
import pathlib as _pathlib

_real = _pathlib.Path(__file__).resolve().parent.parent / "eco-efficient-video-understanding_amd"
__path__ = [str(_real)]
exec(compile((_real / "__init__.py").read_text(), str(_real / "__init__.py"), "exec"))
