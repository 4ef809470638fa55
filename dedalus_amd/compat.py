"""
Run existing d3 scripts unmodified: `import dedalus.public as d3` resolves to dedalus_amd.

    python -m dedalus_amd.compat path/to/script.py [args...]

or, in a script / interpreter,  `import dedalus_amd.compat; dedalus_amd.compat.install()`  before the
first `import dedalus...`.  Only the names this package implements are provided (Cartesian
Fourier x Jacobi problems); anything else raises AttributeError / NotImplementedError loudly.
"""

import runpy
import sys
import types


def install():
    if "dedalus" in sys.modules and getattr(sys.modules["dedalus"], "__dedalus_amd__", False):
        return
    from . import public
    from .core import timesteppers
    from .extras import flow_tools
    pkg = types.ModuleType("dedalus")
    pkg.__dedalus_amd__ = True
    pkg.__path__ = []
    pkg.public = public
    extras = types.ModuleType("dedalus.extras")
    extras.__path__ = []
    extras.flow_tools = flow_tools
    core = types.ModuleType("dedalus.core")
    core.__path__ = []
    core.timesteppers = timesteppers
    pkg.extras, pkg.core = extras, core
    sys.modules.update({"dedalus": pkg, "dedalus.public": public, "dedalus.extras": extras,
                        "dedalus.extras.flow_tools": flow_tools, "dedalus.core": core,
                        "dedalus.core.timesteppers": timesteppers})


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print(__doc__)
        return 2
    install()
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
