"""Importable alias of the hyphenated package directory `audioset-convnext-inf_amd/`.

`import audioset_convnext_inf_amd.pytorch.convnext` mirrors the reference's
`audioset_convnext_inf.pytorch.convnext`; all code lives in `audioset-convnext-inf_amd/`.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "audioset-convnext-inf_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
