"""Import alias: the package directory is `atm-raytracer_amd/` (hyphenated, as the repository
layout prescribes), which Python cannot import by name.  This module makes it importable as
`atm_raytracer_amd` by pointing its package path at that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "atm-raytracer_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
