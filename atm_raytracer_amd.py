"""Import alias: the package directory is `atm-raytracer_amd/` (hyphenated, as the repository
layout prescribes), which Python cannot import by name.  This module makes it importable as
`atm_raytracer_amd` by pointing its package path at that directory; `python -m atm_raytracer_amd ...`
runs the package's command line (atm-raytracer_amd/__main__.py)."""
import os as _os

if __name__ == "__main__":
    import runpy as _runpy
    import sys as _sys

    _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
    import atm_raytracer_amd  # noqa: F401  (registers the alias package)
    _runpy.run_module("atm_raytracer_amd.__main__", run_name="__main__", alter_sys=True)
else:
    __path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "atm-raytracer_amd")]
    __file__ = _os.path.join(__path__[0], "__init__.py")
    with open(__file__) as _f:
        exec(compile(_f.read(), __file__, "exec"))
