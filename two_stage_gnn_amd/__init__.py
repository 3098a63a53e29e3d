"""Importable alias of the package directory ``two-stage-gnn_amd/`` (a hyphen is not a legal module
name).  Everything lives in that directory; this file only points the import system at it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "two-stage-gnn_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
