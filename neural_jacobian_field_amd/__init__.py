"""Importable alias for the ``neural-jacobian-field_amd/`` package directory.

The product package lives in ``neural-jacobian-field_amd/`` (the name the build contract fixes);
a hyphen is not a legal Python identifier, so this stub points its ``__path__`` at that directory
and executes its ``__init__``.  ``import neural_jacobian_field_amd.api`` therefore loads
``neural-jacobian-field_amd/api.py``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "neural-jacobian-field_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
