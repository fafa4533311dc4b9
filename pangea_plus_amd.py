"""Import shim: the package directory is named `pangea-plus_amd/` (not an identifier), so this
module makes it importable as `pangea_plus_amd`."""
import os as _os

__package__ = "pangea_plus_amd"
__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "pangea-plus_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
