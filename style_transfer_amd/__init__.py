"""Import alias: the package sources live in the directory `style-transfer_amd/`, whose
name is not a valid Python identifier.  This module makes `import style_transfer_amd`
resolve to that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "style-transfer_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
