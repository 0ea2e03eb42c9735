"""Importable alias of the ``monocular-depth-estimation_amd`` package.

The package directory carries the project's name, which is not a valid Python
identifier; this alias extends its own search path with that directory so that
``import mde_amd.layers`` resolves to ``monocular-depth-estimation_amd/layers.py``.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "monocular-depth-estimation_amd")
__path__.insert(0, _PKG_DIR)

with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
